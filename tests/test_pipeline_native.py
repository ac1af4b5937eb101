"""CPU: chiron_pipeline_run (csrc/pipeline.cpp: the host side of `chiron call` as one native call -- C++ reader / packer / finisher
threads) against the Python pipeline of chiron_amd/eval.py:evaluation it replaces (chiron_eval.py:304-368, :378-463), behind a NULL
engine: no GPU, the same canned decode on both sides.  Everything the host side produces -- raw/<name>.signal, reference/*_ref.fastq,
result/, segments/, the non-timing lines of meta/ -- must be equal byte for byte: multi-read files, references, a damaged file,
reads of one window, batches that cut reads in two, a partial last batch, DNA / RNA, FASTQ / FASTA, --concise, --no-raw.
(The same comparison behind the real engine is tests/test_gpu_parity.py::test_native_pipeline_equals_the_python_pipeline.)"""
import os

import numpy as np
import pytest

from conftest import ROOT  # noqa: F401

import chiron_amd  # noqa: F401
from chiron_amd import eval as ce, extract as ex


class CannedEngine(object):
    """The engine surface evaluation() uses, returning the decode csrc/pipeline.cpp's null engine returns: row i of EVERY batch decodes
    to counts[i] labels taken from one fixed label stream (an LCG, restated here), path_prob prob[i]."""
    null_engine = True

    def __init__(self, max_batch, segment_len, n_slots=3):
        from chiron_amd.engine import DecodeResult, CompactDecode
        self._DR, self._CD = DecodeResult, CompactDecode
        self.ratio, self.T, self.n_slots, self.max_batch, self.segment_len = 1.0, segment_len, n_slots, max_batch, segment_len
        x = [12345]

        def rnd():
            x[0] = (x[0] * 1664525 + 1013904223) & 0xFFFFFFFF
            return x[0] >> 8
        self.counts = np.asarray([38 + rnd() % 13 for _ in range(max_batch)], dtype=np.int32)
        self.flat = np.asarray([rnd() & 3 for _ in range(int(self.counts.sum()))], dtype=np.uint8)
        self.prob = np.asarray([np.float32(1.0) + np.float32(rnd() % 5000) * np.float32(1e-3) for _ in range(max_batch)], dtype=np.float32).reshape(-1, 1)
        self._pending = [None] * n_slots

    def submit(self, slot, x, seq_len, beam_width=0, want_prob=True, want_logits=False, copy_decoded=True, compact=False):
        assert compact
        self._pending[slot] = int(x.shape[0])

    def submit_pieces(self, slot, pieces, seq_len, beam_width=0, want_prob=True, compact=True):
        self._pending[slot] = int(sum(len(p) for p in pieces))

    def collect(self, slot):
        b, self._pending[slot] = self._pending[slot], None
        nnz = int(self.counts[:b].sum())
        return self._DR(None, np.zeros((b, 1), np.float32), self.prob[:b].copy(), None,
                        self._CD(self.flat[:nnz].copy(), self.counts[:b].copy(), np.asarray([b, int(self.counts.max())], dtype=np.int64)))

    def close(self):
        pass


def _inputs(folder):
    from h5_writer import write_multi_read_fast5
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(31)
    sig = lambda n: rng.randint(200, 1000, size=n).astype(np.int16)
    write_multi_read_fast5(os.path.join(folder, "a_single.fast5"), [("", "id-a", sig(30011), "@x\nACGTTGCA\n+\n!!!!!!!!\n")], chunk=4096)
    write_multi_read_fast5(os.path.join(folder, "b_multi.fast5"), [("read_%03d" % i, "id-b%d" % i, sig(n), ("@y%d\nACG\n+\n!!!\n" % i) if i == 1 else None)
                                                                   for i, n in enumerate((17000, 390, 391, 52001))])
    with open(os.path.join(folder, "c_damaged.fast5"), "wb") as f:
        f.write(b"\x89HDF\r\n\x1a\n" + b"\x00" * 300)
    write_multi_read_fast5(os.path.join(folder, "d_tiny.fast5"), [("", "id-d", sig(7), None)])
    for k in range(6):
        write_multi_read_fast5(os.path.join(folder, "e_more%02d.fast5" % k), [("", "id-e%d" % k, sig(9000 + 1777 * k), None)], chunk=2048)


def _flags(inp, out, **kw):
    class F(object):
        start, segment_len, jump, batch_size = 0, 400, 390, 50
        extension, concise, mode, recursive = "fastq", False, "dna", True
        unit, idname, delimiter, test_number = False, False, "\n", None
        beam, threads, finish_procs, model, no_raw = 0, 3, 0, "null-engine", False
    F.input = F.input_dir = inp
    F.output = F.output_dir = out
    for k, v in kw.items():
        setattr(F, k, v)
    return F


def _tree(root):
    out = {}
    for sub in ("raw", "reference", "result", "segments", "meta"):
        for dp, _, fns in os.walk(os.path.join(root, sub)):
            for n in sorted(fns):
                data = open(os.path.join(dp, n), "rb").read()
                if sub == "meta":          # timings differ run to run: keep the headers and the two setting lines, and the read length
                    lines = data.decode().split("\n")
                    data = "\n".join([lines[0], lines[2], lines[3], lines[4], lines[5]]).encode()
                out[os.path.relpath(os.path.join(dp, n), root)] = data
    return out


@pytest.mark.parametrize("case", [{}, {"extension": "fasta"}, {"mode": "rna"}, {"concise": True}, {"no_raw": True}, {"batch_size": 7, "start": 5}, {"start": 10},
                                  {"segment_len": 300, "jump": 300}, {"segment_len": 400, "jump": 30, "batch_size": 333}])
def test_native_pipeline_equals_the_python_pipeline_behind_a_null_engine(tmp_path, built, case):
    inp = str(tmp_path / "in")
    _inputs(inp)
    trees, results = {}, {}
    for which in ("python", "native"):
        out = str(tmp_path / which)
        F = _flags(inp, out, python_pipeline=(which == "python"), **case)
        ex.prepare_folders(F)
        files = ex.list_fast5(inp)
        eng = CannedEngine(F.batch_size, F.segment_len)
        assert ce.native_pipeline_ok(F, eng, files) == (which == "native")
        results[which] = ce.evaluation(F, engine=eng, fast5_files=files)
        trees[which] = _tree(out)
    assert sorted(results["python"]) == sorted(results["native"]) and len(results["native"]) == 12      # the damaged file is skipped by both
    assert results["native"] == results["python"] and all(isinstance(v, str) for v in results["native"].values())   # name -> consensus string, read back on demand
    assert sorted(trees["python"]) == sorted(trees["native"])
    for name in trees["python"]:
        assert trees["python"][name] == trees["native"][name], name
    st = ce.evaluation.last_native_stats
    assert st["reads"] == st["reads_finished"] == 12 and st["files_failed"] == 1 and st["windows"] > 0
    log = open(os.path.join(str(tmp_path / "native"), "log", "extract.log")).read()
    assert "c_damaged.fast5" in log and "Cannot extract file" in log


@pytest.mark.parametrize("case", [{}, {"extension": "fasta", "batch_size": 9}, {"concise": True, "start": 3}])
def test_native_pipeline_on_a_folder_of_signal_files(tmp_path, built, case):
    """the other input of `chiron call`: a (recursive) folder of `.signal` text files -- what extraction leaves under raw/, or the caller's
    own (chiron_eval.py:277-293, chiron_input.py:527-539): names keep their sub-folder, nothing is written to raw/ or reference/;
    space- and newline-separated files, a float-valued one (pA), an empty one."""
    from chiron_amd import fast5
    inp = tmp_path / "sig"
    (inp / "sub" / "deeper").mkdir(parents=True)
    rng = np.random.RandomState(8)
    fast5.write_signal_text(str(inp / "a.signal"), rng.randint(200, 1000, size=12345).astype(np.float32), "\n")
    fast5.write_signal_text(str(inp / "sub" / "b.signal"), rng.randint(200, 1000, size=4000).astype(np.float32), " ")
    fast5.write_signal_text(str(inp / "sub" / "deeper" / "c.signal"), rng.randint(200, 1000, size=801).astype(np.float32), "\n")
    (inp / "d_float.signal").write_text(" ".join(repr(float(v)) for v in rng.uniform(60.0, 160.0, size=2500)))
    (inp / "e_empty.signal").write_text("")
    (inp / "notes.txt").write_text("not a read")
    trees, results = {}, {}
    for which in ("python", "native"):
        F = _flags(str(inp), str(tmp_path / which), python_pipeline=(which == "python"), **case)
        for sub in ("result", "segments", "meta"):
            os.makedirs(os.path.join(F.output, sub))
        eng = CannedEngine(F.batch_size, F.segment_len)
        results[which] = ce.evaluation(F, engine=eng)
        trees[which] = _tree(F.output)
    assert sorted(results["python"]) == sorted(results["native"]) == ["a.signal", "d_float.signal", "e_empty.signal", "sub/b.signal", "sub/deeper/c.signal"]
    assert ce.evaluation.last_native_stats["reads"] == 5
    py, nat = trees["python"], trees["native"]
    assert sorted(py) == sorted(nat) and all(py[k] == nat[k] for k in py), [k for k in py if py.get(k) != nat.get(k)]
    assert "result/sub/deeper/c." + case.get("extension", "fastq") in nat


def test_native_pipeline_refuses_a_signal_file_that_is_not_numbers(tmp_path, built):
    from chiron_amd import _lib
    inp = tmp_path / "sig"
    inp.mkdir()
    (inp / "bad.signal").write_text("487 421 oops 433")
    F = _flags(str(inp), str(tmp_path / "out"))
    for sub in ("result", "segments", "meta"):
        os.makedirs(os.path.join(F.output, sub))
    with pytest.raises(_lib.ChironError) as err:
        ce.evaluation(F, engine=CannedEngine(F.batch_size, F.segment_len))
    assert "bad.signal" in str(err.value)


def test_native_pipeline_reports_a_failed_write(tmp_path, built):
    """result/ missing: the finisher cannot write -- the call fails with a status and a message instead of dropping reads silently"""
    from chiron_amd import _lib
    inp = str(tmp_path / "in")
    _inputs(inp)
    F = _flags(inp, str(tmp_path / "out"))
    ex.prepare_folders(F)
    for sub in ("segments", "meta"):
        os.makedirs(os.path.join(F.output, sub))
    eng = CannedEngine(F.batch_size, F.segment_len)
    with pytest.raises(_lib.ChironError):
        ce.run_native_pipeline(F, eng, ex.list_fast5(inp), 2)
