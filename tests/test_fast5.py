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
