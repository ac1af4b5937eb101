#!/usr/bin/env python3
"""Complete the example_dna fixture (runs ONLY in the build container, where /root/reference exists).

make_golden.py copied read1.fast5 and raw/read1.signal; BASELINE configs[0] is `chiron call` on all FIVE example
reads, so this adds read2..5.fast5 (data files of the reference's example, 3.7 MB) and, instead of the four remaining
raw/readN.signal text files (4 MB of decimal text), a digest of every raw/readN.signal the reference checked in:
sample count, SHA-256 of the samples as little-endian int16, first and last five values.  The HDF5 reader
(chiron_amd/fast5.py) must reproduce those from the fast5 files (tests/test_fast5.py).

    python tests/golden/make_example_fixture.py
"""
import hashlib
import json
import os
import shutil

import numpy as np

EX = "/root/reference/chiron/example_data/DNA"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "example_dna")


def main():
    digest = {}
    for i in range(1, 6):
        name = "read%d" % i
        dst = os.path.join(DST, name + ".fast5")
        shutil.copyfile(os.path.join(EX, name + ".fast5"), dst)
        os.chmod(dst, 0o644)
        vals = np.asarray(open(os.path.join(EX, "output", "raw", name + ".signal")).read().split(), dtype=np.float64)
        assert np.all(vals == np.rint(vals)) and np.abs(vals).max() < 32768
        a = vals.astype("<i2")
        seq = open(os.path.join(EX, "output", "result", name + ".fastq")).read().split("\n")[1]
        digest[name] = {"samples": int(a.size), "sha256_int16le": hashlib.sha256(a.tobytes()).hexdigest(),
                        "head": a[:5].tolist(), "tail": a[-5:].tolist(), "windows_L400_J390": int(-(-a.size // 390)),
                        "reference_consensus_len": len(seq)}
    with open(os.path.join(DST, "raw_digest.json"), "w") as f:
        json.dump(digest, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(digest, indent=1))


if __name__ == "__main__":
    main()
