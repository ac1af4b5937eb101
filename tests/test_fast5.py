"""CPU: the h5py-free fast5 reader and extract step (rows H1 of SURVEY.md 8a) against the reference's
own example file: read1.fast5 must yield exactly the samples of the reference's raw/read1.signal."""
import os

import numpy as np
import pytest

from chiron_amd import extract, fast5, signal_io
from conftest import GOLDEN

F5 = os.path.join(GOLDEN, "example_dna", "read1.fast5")
SIG = os.path.join(GOLDEN, "example_dna", "raw", "read1.signal")


def test_signal_matches_reference_extract():
    recs = fast5.read_fast5(F5)
    assert len(recs) == 1
    sig = recs[0]["signal"]
    assert sig.dtype == np.int16 and sig.shape == (62461,)
    assert np.array_equal(sig.astype(np.float32), signal_io.read_signal(SIG))
    ch = recs[0]["channel"]
    assert ch["digitisation"] == 8192.0 and ch["sampling_rate"] == 4000.0 and abs(ch["range"] - 1485.56) < 1e-2
    assert np.array_equal(fast5.read_raw_signal(F5), sig)
    # chiron_input.read_signal_fast5 equivalent
    assert np.array_equal(signal_io.read_signal_fast5(F5).astype(np.float32), sig.astype(np.float32))


def test_read_data_for_eval_accepts_fast5_and_reverses():
    ds = signal_io.read_data_for_eval(F5, 0, 390, 400)
    ref = signal_io.read_data_for_eval(SIG, 0, 390, 400)
    assert np.array_equal(ds.event, ref.event) and np.array_equal(ds.event_length, ref.event_length)
    rv = signal_io.read_data_for_eval(F5, 0, 390, 400, reverse_fast5=True)     # chiron_input.py:269-272
    assert rv.event[0, 0] == ref.event[-1][ref.event_length[-1] - 1]


def test_extract_writes_signal_files(tmp_path):
    import shutil
    inp = tmp_path / "in" / "sub"
    inp.mkdir(parents=True)
    shutil.copy(F5, str(inp / "read1.fast5"))
    (inp / "broken.fast5").write_bytes(b"not hdf5 at all")
    (inp / "readme.txt").write_text("x")

    class F(object):
        input_dir, output_dir = str(tmp_path / "in"), str(tmp_path / "out")
        mode, unit, recursive, idname, delimiter, threads, test_number = "dna", False, True, False, "\n", 1, None
    n = extract.extract(F)
    assert n == 1
    out = open(os.path.join(F.output_dir, "raw", "read1.signal")).read()
    assert out.split("\n")[:5] == ["487", "421", "433", "438", "452"] and out.count("\n") == 62460
    assert np.array_equal(signal_io.read_signal(os.path.join(F.output_dir, "raw", "read1.signal")),
                          signal_io.read_signal(SIG))
    log = open(os.path.join(F.output_dir, "log", "extract.log")).read()
    assert "broken.fast5" in log                                  # unreadable file logged and skipped
    for d in ("raw", "reference", "log"):
        assert os.path.isdir(os.path.join(F.output_dir, d))
    # rna mode reverses the signal (extract_sig_ref.py:165); unit converts to pA (:153-158)
    F.mode, F.output_dir = "rna", str(tmp_path / "out2")
    extract.extract(F)
    r = signal_io.read_signal(os.path.join(F.output_dir, "raw", "read1.signal"))
    assert np.array_equal(r, signal_io.read_signal(SIG)[::-1])
    recs = extract.extract_file(F5, "dna", unit=True)
    assert abs(recs[0][1][0] - (487 + 18.0) * 1485.56 / 8192.0) < 1e-3


def test_not_hdf5_raises():
    with pytest.raises(fast5.Fast5FormatError):
        fast5.read_fast5(__file__)


def test_multi_read_fast5_extract_file_v2(tmp_path):
    """Multi-read files (extract_sig_ref.py:178-193 extract_file_v2): one record per top-level read group, read_id
    from the Raw group, reference from that read's own Analyses tree.  The reference ships no multi-read example,
    so the file comes from tests/h5_writer.py (contiguous and chunked+deflate signal storage)."""
    from h5_writer import write_multi_read_fast5
    from chiron_amd import extract
    rng = np.random.RandomState(4)
    reads = [("read_%04d" % i, "id-%d-abc" % i, rng.randint(200, 1000, size=n).astype(np.int16),
              ("@x\nACGT%d\n+\n!!!!!\n" % i) if i != 1 else None) for i, n in enumerate((5000, 1234, 40001))]
    for chunk in (None, 4096):
        p = str(tmp_path / ("multi_%s.fast5" % chunk))
        write_multi_read_fast5(p, reads, chunk=chunk)
        recs = fast5.read_fast5(p)
        assert [r["suffix"] for r in recs] == [r[0] for r in reads]
        for rec, (name, rid, sig, fq) in zip(recs, reads):
            assert rec["read_id"] == rid and rec["signal"].dtype == np.int16 and np.array_equal(rec["signal"], sig)
            assert rec["fastq"] == (fq or "")
    # the extract step writes one .signal per read, named by read id as the reference does for multi-read input
    out = extract.extract_file(p, mode="dna")
    assert len(out) == 3
    rna = extract.extract_file(p, mode="rna")
    assert np.array_equal(rna[0][1], reads[0][2][::-1]) and rna[0][3] == "id-0-abc" and out[2][2].startswith("@x")


def test_all_five_example_fast5_reproduce_the_reference_raw_signals():
    """BASELINE configs[0] input: chiron/example_data/DNA/read{1..5}.fast5.  The reference checked in what its own
    extraction produced (output/raw/readN.signal); tests/golden/example_dna/raw_digest.json holds sample count, SHA-256
    and ends of each.  The h5py-free reader must give exactly those samples for all five files."""
    import hashlib
    import json
    ex = os.path.join(GOLDEN, "example_dna")
    digest = json.load(open(os.path.join(ex, "raw_digest.json")))
    assert sorted(digest) == ["read%d" % i for i in range(1, 6)]
    total_windows = 0
    for name, d in digest.items():
        recs = fast5.read_fast5(os.path.join(ex, name + ".fast5"))
        assert len(recs) == 1
        sig = np.asarray(recs[0]["signal"])
        assert sig.dtype == np.int16 and sig.size == d["samples"]
        assert sig[:5].tolist() == d["head"] and sig[-5:].tolist() == d["tail"]
        assert hashlib.sha256(sig.astype("<i2").tobytes()).hexdigest() == d["sha256_int16le"]
        ds = signal_io.read_data_for_eval(os.path.join(ex, name + ".fast5"), 0, 390, 400)
        assert ds.reads_n == d["windows_L400_J390"]
        total_windows += ds.reads_n
    assert total_windows == 2688              # 2688 windows -> 27 batches of 100, the last wrap-padded by 12


def test_extraction_shards_across_ranks(tmp_path):
    """Sharded `chiron call` (one process per GPU): rank r extracts files r, r + world, ... of the sorted list with its
    own pool; together the ranks produce exactly what a single process does (extract_sig_ref.py:58-60,81 spread over
    ranks).  test_number cuts the GLOBAL list before sharding."""
    import filecmp
    import shutil
    from h5_writer import write_multi_read_fast5
    inp = tmp_path / "in"
    (inp / "deep").mkdir(parents=True)
    rng = np.random.RandomState(9)
    for k in range(7):
        reads = [("read_%d" % k, "id%d" % k, rng.randint(200, 1000, size=3000 + 17 * k).astype(np.int16), None)]
        write_multi_read_fast5(str((inp / "deep" if k % 3 == 0 else inp) / ("f%02d.fast5" % k)), reads, chunk=1024 if k % 2 else None)
    shutil.copy(F5, str(inp / "z_example.fast5"))

    def flags(out, test_number=None):
        import argparse                       # a Namespace like the CLI's: it travels to the pool workers
        return argparse.Namespace(input_dir=str(inp), output_dir=str(out), mode="dna", unit=False, recursive=True, idname=False,
                                  delimiter="\n", threads=2, test_number=test_number)
    one = flags(tmp_path / "one")
    assert extract.extract(one) == 8
    two = tmp_path / "two"
    counts = [extract.extract(flags(two), rank=r, world=2) for r in range(2)]
    assert sum(counts) == 8 and min(counts) == 4
    a, b = sorted(os.listdir(str(tmp_path / "one" / "raw"))), sorted(os.listdir(str(two / "raw")))
    assert a == b and len(a) == 8
    for n in a:
        assert filecmp.cmp(str(tmp_path / "one" / "raw" / n), str(two / "raw" / n), shallow=False)
    assert sorted(os.listdir(str(two / "log"))) == ["extract.rank0.log", "extract.rank1.log"]
    three = tmp_path / "three"
    assert sum(extract.extract(flags(three, test_number=3), rank=r, world=2) for r in range(2)) == 3
    assert len(extract.list_fast5(str(inp), True)) == 8 and len(extract.list_fast5(str(inp), False)) == 5


def test_native_reader_equals_the_python_reader_on_every_fixture(tmp_path):
    """csrc/fast5.cpp (chiron_fast5_*: the reader of the direct `chiron call` path) against fast5.py on
    the reference's five example files (old-style groups, chunked + deflate int16 signal: sample counts and SHA-256 of the
    reference's own raw/readN.signal), on multi-read files written by tests/h5_writer.py (contiguous and chunked +
    deflate, with and without a reference FASTQ), reversed (RNA), and on damaged input (reason reported, nothing
    crashes)."""
    import hashlib
    import json
    import h5_writer
    ex = os.path.join(GOLDEN, "example_dna")
    digest = json.load(open(os.path.join(ex, "raw_digest.json")))
    for name, d in digest.items():
        p = os.path.join(ex, name + ".fast5")
        py, nat = fast5.read_fast5(p), fast5.read_fast5_native(p)
        assert len(py) == len(nat) == 1 and nat[0]["suffix"] == "" and nat[0]["signal"].dtype == np.float32
        assert np.array_equal(np.asarray(py[0]["signal"]).astype(np.float32), nat[0]["signal"])
        assert nat[0]["signal"].size == d["samples"]
        assert hashlib.sha256(nat[0]["signal"].astype("<i2").tobytes()).hexdigest() == d["sha256_int16le"]
        assert py[0]["read_id"] == nat[0]["read_id"] and py[0]["fastq"] == nat[0]["fastq"]
        assert np.array_equal(fast5.read_fast5_native(p, reverse=True)[0]["signal"], nat[0]["signal"][::-1])
    rng = np.random.RandomState(0)
    for chunk in (None, 1000, 7):
        reads = [("read_%d" % k, "id-%d" % k, rng.randint(-300, 1200, size=n).astype(np.int16), ("@x\nACGT\n+\n!!!!" if k % 2 else None))
                 for k, n in enumerate((5000, 1, 12345))]
        p = str(tmp_path / ("multi_%s.fast5" % chunk))
        h5_writer.write_multi_read_fast5(p, reads, chunk=chunk)
        py, nat = fast5.read_fast5(p), fast5.read_fast5_native(p)
        assert [r["suffix"] for r in nat] == ["read_0", "read_1", "read_2"] == [r["suffix"] for r in py]
        for a, b, (_, rid, sig, fq) in zip(py, nat, reads):
            assert np.array_equal(b["signal"], sig.astype(np.float32)) and np.array_equal(np.asarray(a["signal"]), sig)
            assert b["read_id"] == rid == a["read_id"] and b["fastq"] == (fq or "") == a["fastq"]
    # damaged input: an error with a reason, never a crash
    good = open(os.path.join(ex, "read1.fast5"), "rb").read()
    bad = tmp_path / "bad.fast5"
    for blob in (b"", b"not hdf5 at all", good[:100], good[:5000], good[:len(good) // 2],
                 good[:8] + bytes([7]) + good[9:], good[:2048] + bytes(len(good) - 2048)):
        bad.write_bytes(blob)
        with pytest.raises(fast5.Fast5FormatError):
            fast5.read_fast5_native(str(bad))
    with pytest.raises(fast5.Fast5FormatError):
        fast5.read_fast5_native(str(tmp_path / "missing.fast5"))
    # a flipped byte inside a deflate stream
    h5_writer.write_multi_read_fast5(str(bad), [("r", "i", rng.randint(0, 900, size=4000).astype(np.int16), None)], chunk=4000)
    blob = bytearray(bad.read_bytes())
    blob[300] ^= 0xFF
    bad.write_bytes(bytes(blob))
    try:
        got = fast5.read_fast5_native(str(bad))
        assert got[0]["signal"].size == 4000          # the flip hit padding: still a valid file
    except fast5.Fast5FormatError:
        pass


def test_signal_text_writer_and_extract_records(tmp_path):
    """chiron_write_signal_text == delimiter.join(str(v) for v in raw) (extract_sig_ref.py:122-123) for integer samples,
    negative values and the empty signal included; a non-integer sample is refused.  extract.extract_records returns the
    samples it wrote (what the direct path windows) and the text parses back to them."""
    sig = np.array([487, -3, 0, 32767, -32768, 12], dtype=np.int16)
    for delim in ("\n", " "):
        p = str(tmp_path / "a.signal")
        fast5.write_signal_text(p, sig.astype(np.float32), delim)
        assert open(p).read() == delim.join(str(v) for v in sig.tolist())
    fast5.write_signal_text(p, np.zeros(0, dtype=np.float32))
    assert open(p).read() == ""
    from chiron_amd._lib import ChironError
    with pytest.raises(ChironError):
        fast5.write_signal_text(p, np.array([1.5], dtype=np.float32))

    class F(object):
        mode, unit, idname, delimiter = "dna", False, False, "\n"
        raw_folder, ref_folder = str(tmp_path / "raw"), str(tmp_path / "reference")
    os.makedirs(F.raw_folder)
    os.makedirs(F.ref_folder)
    recs = extract.extract_records(F5, F)
    assert len(recs) == 1 and recs[0][0] == "read1" and recs[0][1].dtype == np.float32
    assert np.array_equal(signal_io.read_signal(os.path.join(F.raw_folder, "read1.signal")), recs[0][1])
    assert np.array_equal(recs[0][1], signal_io.read_signal(SIG))
    F.mode = "rna"
    assert np.array_equal(extract.extract_records(F5, F)[0][1], recs[0][1][::-1])
    # `chiron call --no-raw` (opt-in deviation, DESIGN appendix D Q28): the samples come back, raw/<name>.signal is not written; the
    # two-pass path (--via-signal-files) reads that file back, so entry.evaluation switches the option off there
    F.mode, F.no_raw = "dna", True
    os.remove(os.path.join(F.raw_folder, "read1.signal"))
    again = extract.extract_records(F5, F)
    assert np.array_equal(again[0][1], recs[0][1]) and os.listdir(F.raw_folder) == []


def test_native_reader_with_and_without_libdeflate(tmp_path):
    """The deflate filter goes through libdeflate when the box has libdeflate.so.0 (looked up with dlopen, once per process)
    and through zlib otherwise or with CHIRON_NO_LIBDEFLATE=1: a child process with the switch set must read the same samples
    as this process (chunked + deflate int16, several chunk sizes, a chunk that does not fill its last piece)."""
    import hashlib
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import h5_writer
    from chiron_amd import fast5 as f5
    rng = np.random.RandomState(4)
    paths = []
    for i, (n, chunk) in enumerate(((100000, 20000), (12345, 4096), (70001, 70001))):
        sig = (rng.randn(n) * 80 + 500).astype(np.int16)
        pth = str(tmp_path / ("r%d.fast5" % i))
        h5_writer.write_multi_read_fast5(pth, [("", "id-%d" % i, sig, None)], chunk=chunk)
        paths.append((pth, hashlib.sha256(sig.astype(np.float32).tobytes()).hexdigest()))
    code = ("import sys, hashlib, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from chiron_amd import fast5 as f5\n"
            "for p in sys.argv[1:]:\n"
            "    recs = f5.read_fast5_native(p)\n"
            "    print(hashlib.sha256(np.asarray(recs[0]['signal'], dtype=np.float32).tobytes()).hexdigest())\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    want = [h for _, h in paths]
    for env_extra in ({}, {"CHIRON_NO_LIBDEFLATE": "1"}):
        env = dict(os.environ)
        env.update(env_extra)
        out = subprocess.run([sys.executable, "-c", code] + [p for p, _ in paths], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.split() == want, env_extra


@pytest.mark.timeout(120)
def test_crafted_damage_is_reported_never_a_crash_or_a_hang(tmp_path):
    """Files damaged in ways a truncation never produces (round-3 advisor findings on csrc/fast5.cpp): an element size of 0 in
    front of the shuffle filter (was: integer division by zero, SIGFPE), a dimension of 2^63 (was: the byte count wrapped, the
    size guards passed and chiron_fast5_signal wrote past `cap`), a chunk offset near 2^64 (s + cbytes wrapped), a chunk B-tree
    whose children point back at the node (was: used^32 visits), a global-heap object size that makes the scan stand still.
    Both readers must answer with Fast5FormatError (the native one through CHIRON_ERR_INVALID / _OVERFLOW and a reason)."""
    import ctypes as C
    import struct
    import h5_writer
    from chiron_amd import _lib
    rng = np.random.RandomState(11)
    sig = rng.randint(0, 900, size=3000).astype(np.int16)

    def both_fail(path):
        with pytest.raises(fast5.Fast5FormatError):
            fast5.read_fast5_native(path)
        with pytest.raises((fast5.Fast5FormatError, ValueError, struct.error, IndexError, zlib_error)):
            fast5.read_fast5(path)
    import zlib
    zlib_error = zlib.error

    def patched(name, chunk, edit):
        p = str(tmp_path / name)
        h5_writer.write_multi_read_fast5(p, [("", "id", sig, None)], chunk=chunk)
        blob = bytearray(open(p, "rb").read())
        edit(blob)
        open(p, "wb").write(bytes(blob))
        return p

    dt16 = struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 2)

    # (1) filter id deflate -> shuffle, element size 2 -> 0
    def size0(blob):
        i = blob.index(dt16)
        blob[i + 4:i + 8] = struct.pack("<I", 0)
        j = blob.index(struct.pack("<HHHH", 1, 0, 1, 1))
        blob[j:j + 2] = struct.pack("<H", 2)
    both_fail(patched("size0.fast5", 1000, size0))
    # an element size the conversion has no case for
    both_fail(patched("size3.fast5", None, lambda blob: blob.__setitem__(slice(blob.index(dt16) + 4, blob.index(dt16) + 8), struct.pack("<I", 3))))

    # (2) dimension 2^63 on a contiguous dataset, and the C entry point called directly with a small capacity
    def huge_dim(blob):
        i = blob.index(struct.pack("<BBB5xQ", 1, 1, 0, 3000))
        blob[i + 8:i + 16] = struct.pack("<Q", 1 << 63)
    p = patched("dim63.fast5", None, huge_dim)
    both_fail(p)
    lib = _lib.load()
    h = C.c_void_p()
    st = lib.chiron_fast5_open(p.encode(), C.byref(h))
    if st == _lib.OK:                                   # open may already refuse the file; if not, the read must
        out = np.zeros(16, dtype=np.float32)
        assert lib.chiron_fast5_signal(h, 0, out.ctypes.data_as(C.c_void_p), 16, 0) != _lib.OK
        lib.chiron_fast5_close(h)
    for dim in ((1 << 62) + 5, (1 << 61), 1 << 41):
        def f(blob, dim=dim):
            i = blob.index(struct.pack("<BBB5xQ", 1, 1, 0, 3000))
            blob[i + 8:i + 16] = struct.pack("<Q", dim)
        both_fail(patched("dim.fast5", None, f))

    # (3) a chunk whose offset times the element size is just below 2^64
    def far_chunk(blob):
        t = blob.index(b"TREE")
        blob[t + 24 + 8:t + 24 + 16] = struct.pack("<Q", (1 << 63) - 4)
    pth = patched("far.fast5", 1000, far_chunk)
    got = fast5.read_fast5_native(pth)                 # the chunk is ignored (it lies outside the dataset); nothing is overwritten
    assert got[0]["signal"].size == 3000 and np.array_equal(got[0]["signal"][1000:], sig[1000:].astype(np.float32))

    # (4) a chunk B-tree of level 1 whose children are the node itself
    def loop(blob):
        t = blob.index(b"TREE")
        blob[t + 5] = 1
        for k in range(3):
            blob[t + 24 + 32 * k + 24:t + 24 + 32 * k + 32] = struct.pack("<Q", t)
    both_fail(patched("loop.fast5", 1000, loop))

    # (5) global heap: read_id as a variable-length string whose heap holds an object of size 2^64 - 16 before the wanted one
    b = h5_writer.H5Builder()
    heap = b"GCOL" + struct.pack("<B3xQ", 1, 4096)
    heap += struct.pack("<HHIQ", 7, 0, 0, 0xFFFFFFFFFFFFFFF0) + b"x" * 16
    heap += struct.pack("<HHIQ", 1, 0, 0, 8) + b"read-one"
    heap += b"\x00" * (4096 - len(heap))
    gaddr = b._put(heap)
    vdt = struct.pack("<BBBBI", 0x19, 0x01, 0, 0, 16) + struct.pack("<BBBBI", 0x10, 0, 0, 0, 1) + struct.pack("<HH", 0, 8)
    nm = b"read_id\x00"
    body = struct.pack("<BxHHH", 1, len(nm), len(vdt), 8) + h5_writer._pad8(nm) + h5_writer._pad8(vdt) + h5_writer._space([]) \
        + struct.pack("<IQI", 8, gaddr, 1)
    raw = b.group({"Signal": b.dataset_int16(sig)}, attrs=[h5_writer._msg(0x0C, body)])
    p = str(tmp_path / "heap.fast5")
    b.finish(b.group({"": b.group({"Raw": raw})}), p)
    both_fail(p)
    # the same file with a sane first object reads, and its read_id comes from the heap
    blob = bytearray(open(p, "rb").read())
    i = blob.index(struct.pack("<Q", 0xFFFFFFFFFFFFFFF0))
    blob[i:i + 8] = struct.pack("<Q", 16)
    open(p, "wb").write(bytes(blob))
    assert fast5.read_fast5_native(p)[0]["read_id"] == "read-one" == fast5.read_fast5(p)[0]["read_id"]


def test_reference_rna_example_fast5_reproduce_their_digest():
    """chiron/example_data/RNA/*.fast5 -- the reference's five RNA example reads (it ships no outputs for them): both readers give
    the pinned read_id, sample count and SHA-256 of the int16 signal in acquisition order (tests/golden/example_rna/raw_digest.json,
    written by make_example_rna_fixture.py, which also decodes every file WITHOUT any HDF5 structure -- zlib streams found by brute
    force -- and requires the same samples); reverse = the RNA orientation `--mode rna` feeds the network (extract_sig_ref.py:165)."""
    import hashlib
    import json
    ex = os.path.join(GOLDEN, "example_rna")
    digest = json.load(open(os.path.join(ex, "raw_digest.json")))
    assert len(digest) == 5
    for name, d in digest.items():
        p = os.path.join(ex, name)
        py, nat = fast5.read_fast5(p), fast5.read_fast5_native(p)
        assert len(py) == len(nat) == 1 and nat[0]["suffix"] == ""
        for rec, sig in ((py[0], np.asarray(py[0]["signal"])), (nat[0], nat[0]["signal"])):
            assert sig.size == d["samples"] and rec["read_id"] == d["read_id"] and rec["fastq"] == ""
            assert hashlib.sha256(sig.astype("<i2").tobytes()).hexdigest() == d["sha256_int16le"]
            assert sig[:5].tolist() == d["head"] and sig[-5:].tolist() == d["tail"] and sig.min() == d["min"] and sig.max() == d["max"]
        assert np.array_equal(fast5.read_fast5_native(p, reverse=True)[0]["signal"], nat[0]["signal"][::-1])
        ds = signal_io.read_data_for_eval(p, 0, step=1900, seg_length=2000, reverse_fast5=True)
        assert ds.reads_n == d["windows_L2000_J1900"]
        assert np.array_equal(np.asarray(ds.event[0][:5], dtype=np.float32), np.asarray(d["tail"][::-1], dtype=np.float32))


def test_same_basename_in_two_subfolders_is_one_read(tmp_path):
    """extract_sig_ref.py:62-79 walks recursively and names the output by the file stem: of two files with one basename the later
    overwrites the earlier.  extract.unique_read_files keeps exactly that one for the direct path (which would otherwise merge both
    files' windows under one read name)."""
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    files = []
    for d in (a, b):
        for n in ("x.fast5", "y.fast5" if d is a else "z.fast5"):
            (d / n).write_bytes(b"")
            files.append(str(d / n))
    files.sort()
    assert extract.list_fast5(str(tmp_path), True) == files
    keep, dropped = extract.unique_read_files(files)
    assert keep == [str(a / "y.fast5"), str(b / "x.fast5"), str(b / "z.fast5")]
    assert dropped == [(str(a / "x.fast5"), str(b / "x.fast5"))]
    assert extract.unique_read_files(keep) == (keep, [])
