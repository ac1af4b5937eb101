#!/usr/bin/env python3
"""Pin the reference's RNA example input (runs ONLY in the build container, where /root/reference exists).

chiron/example_data/RNA/ holds five single-read fast5 files and no outputs.  This script copies the five data files to
tests/golden/example_rna/ (168 KB) and writes raw_digest.json: per file the read_id attribute, the sample count, the SHA-256
of the raw signal as little-endian int16 in ACQUISITION order (what extract_sig_ref.py:149-163 reads before `--mode rna`
reverses it, :165), head / tail values and the window count at the RNA preset (segment_len 2000, jump 1900, entry.py:26-27)
and at BASELINE configs[2]'s 500 / 490.

h5py is not installed here, so the samples come from this repo's Python HDF5 reader (chiron_amd/fast5.py).  To keep the digest
from pinning that reader to itself, every file is ALSO decoded without any HDF5 structure at all: every zlib stream found by
brute force in the file image (a 0x78 byte at which zlib.decompressobj succeeds and yields a whole number of int16 samples),
concatenated in file order, must be the reader's signal followed by the zero padding of the last chunk.

    python tests/golden/make_example_rna_fixture.py
"""
import glob
import hashlib
import json
import os
import shutil
import sys
import zlib

import numpy as np

EX = "/root/reference/chiron/example_data/RNA"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "example_rna")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def brute_force_streams(blob):
    """every zlib stream of at least 1 KB of output in the image, in file order: [(offset, bytes)]"""
    out, i = [], 0
    while True:
        i = blob.find(b"\x78", i)
        if i < 0:
            return out
        d = zlib.decompressobj()
        try:
            data = d.decompress(blob[i:])
            if d.eof and len(data) >= 1024 and len(data) % 2 == 0:
                out.append((i, data))
                i += len(blob[i:]) - len(d.unused_data)
                continue
        except zlib.error:
            pass
        i += 1


def main():
    from chiron_amd import fast5
    os.makedirs(DST, exist_ok=True)
    digest = {}
    for src in sorted(glob.glob(os.path.join(EX, "*.fast5"))):
        name = os.path.basename(src)
        recs = fast5.read_fast5(src)
        assert len(recs) == 1
        sig = np.asarray(recs[0]["signal"])
        assert sig.dtype == np.dtype("<i2")
        streams = brute_force_streams(open(src, "rb").read())
        cat = np.frombuffer(b"".join(d for _, d in streams), dtype="<i2")
        assert cat.size >= sig.size and np.array_equal(cat[:sig.size], sig) and not cat[sig.size:].any(), name
        dst = os.path.join(DST, name)
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        digest[name] = {"read_id": recs[0]["read_id"], "samples": int(sig.size),
                        "sha256_int16le": hashlib.sha256(sig.astype("<i2").tobytes()).hexdigest(),
                        "head": sig[:5].tolist(), "tail": sig[-5:].tolist(), "min": int(sig.min()), "max": int(sig.max()),
                        "deflate_streams_found_by_brute_force": len(streams),
                        "windows_L2000_J1900": int(-(-sig.size // 1900)), "windows_L500_J490": int(-(-sig.size // 490))}
    with open(os.path.join(DST, "raw_digest.json"), "w") as f:
        json.dump(digest, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(digest, indent=1))


if __name__ == "__main__":
    main()
