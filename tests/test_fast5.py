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
