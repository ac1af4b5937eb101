"""Serving surface (chiron_amd/serve.py): the PREDICT signature of export_test.py:103-112 over the local wire and
the chiron_client.py flow on top of it.  CPU tests run against a stand-in engine whose logits are a deterministic
function of the input and whose decode is the oracle's greedy CTC; the GPU test (test_gpu_parity.py) uses the
real engine."""
import os
import threading
from collections import namedtuple

import numpy as np
import pytest

from conftest import GOLDEN
from chiron_amd import serve, eval as ce, signal_io
from chiron_amd.engine import SparseTensor, DecodeResult, seq_len_for_engine
from oracle import ctc_oracle


class StandInEngine(object):
    """Same call surface as chiron_amd.Engine.infer; logits from a fixed random projection of window statistics."""

    def __init__(self, segment_len=400, max_batch=7, n_slots=2):
        self.segment_len, self.max_batch, self.n_slots = segment_len, max_batch, n_slots
        self.T, self.ratio = segment_len // 4, 4.0
        self._w = np.random.RandomState(3).randn(4, 5).astype(np.float32) * 3
        self.calls = []
        self._lock = threading.Lock()

    def infer(self, x, seq_len, beam_width=0, want_prob=True, want_logits=False, slot=0):
        assert x.shape[0] <= self.max_batch and x.shape[1] == self.segment_len and 0 <= slot < self.n_slots
        with self._lock:
            self.calls.append((x.shape[0], slot))
        f = x.reshape(x.shape[0], self.T, 4)
        logits = (f - f.mean(axis=(1, 2), keepdims=True)) / (f.std(axis=(1, 2), keepdims=True) + 1) @ self._w
        rows, nsl = ctc_oracle.greedy_decode(logits, seq_len)
        idx, val, shape = ctc_oracle.rows_to_sparse(rows, x.shape[0])
        return DecodeResult(SparseTensor(idx, val, np.asarray(shape, np.int64)), np.asarray(nsl, np.float32).reshape(-1, 1),
                            ctc_oracle.path_prob(logits).astype(np.float32).reshape(-1, 1), logits if want_logits else None)


@pytest.fixture()
def served():
    eng = StandInEngine()
    with serve.PredictServer(eng, ("127.0.0.1", 0)) as srv:
        yield eng, srv


def test_signature_roundtrip_and_batch_splitting(served):
    eng, srv = served
    rng = np.random.RandomState(0)
    x = rng.randn(17, 400).astype(np.float32)                 # 17 rows > max_batch 7: 3 engine batches
    sl = rng.randint(0, 401, size=17)
    sl[:2] = [0, 400]
    with serve.PredictClient(srv.address, srv.authkey) as c:
        sig = c.signature()
        assert tuple(sig["inputs"]) == ("x", "seq_len") and sig["segment_len"] == 400 and sig["ratio"] == 4.0
        assert set(sig["outputs"]) == {"indices", "values", "dense_shape", "logits", "prob_logits", "log_prob"}
        out = c.predict(x, sl)
    assert [n for n, _ in eng.calls] == [7, 7, 3]
    # same as one in-process decode of the whole batch with the server-side seq_len rounding (export_test.py:34)
    big = StandInEngine(max_batch=64)
    ref = big.infer(x, seq_len_for_engine(sl, 4.0), want_logits=True)
    assert np.array_equal(out["indices"], ref.decoded.indices) and np.array_equal(out["values"], ref.decoded.values)
    assert out["dense_shape"][0] == 17 and out["dense_shape"][1] == ref.decoded.dense_shape[1]
    np.testing.assert_allclose(out["logits"], ref.logits, rtol=0, atol=0)
    np.testing.assert_array_equal(out["prob_logits"], ref.prob_logits)
    np.testing.assert_array_equal(out["log_prob"], ref.log_prob)
    assert out["indices"].dtype == np.int64 and out["log_prob"].shape == (17, 1)


def test_errors_travel_to_the_caller_and_the_server_survives(served):
    eng, srv = served
    with serve.PredictClient(srv.address, srv.authkey) as c:
        with pytest.raises(serve.PredictError, match="x must be"):
            c.predict(np.zeros((2, 399), np.float32), np.zeros(2, np.int32))
        with pytest.raises(serve.PredictError, match="seq_len has"):
            c.predict(np.zeros((2, 400), np.float32), np.zeros(3, np.int32))
        out = c.predict(np.zeros((0, 400), np.float32), np.zeros(0, np.int32))      # empty request
        assert out["values"].shape == (0,) and out["dense_shape"].tolist() == [0, 0]
        assert c.predict(np.ones((1, 400), np.float32), [400])["log_prob"].shape == (1, 1)
    with pytest.raises(Exception):                                                    # wrong key: handshake refused
        serve.PredictClient(srv.address, b"nope").signature()
    with serve.PredictClient(srv.address, srv.authkey) as c:                                       # server still answers
        assert c.signature()["T"] == 100


def test_concurrent_requests_use_the_engine_slots(served):
    eng, srv = served
    rng = np.random.RandomState(1)
    xs = [rng.randn(5, 400).astype(np.float32) for _ in range(12)]
    with serve.PredictClient(srv.address, srv.authkey, concurrency=4) as c:
        futs = [c.predict_future(x, np.full(5, 400), want_logits=True) for x in xs]
        outs = [f.result() for f in futs]
    one = StandInEngine(max_batch=64)
    for x, o in zip(xs, outs):
        assert np.array_equal(o["values"], one.infer(x, np.full(5, 100, np.int32)).decoded.values)
    assert {s for _, s in eng.calls} <= {0, 1} and len(eng.calls) == 12


def test_client_flow_writes_the_chiron_eval_output_tree(served, tmp_path):
    """chiron_client.do_inference on the reference's example signal: jump 30 -> `simple` assembly kernel, zero-padded
    last batch, result/segments/meta written; equals the same pipeline run in-process."""
    eng, srv = served
    inp = tmp_path / "in"
    inp.mkdir()
    sig = signal_io.read_signal(os.path.join(GOLDEN, "example_dna", "raw", "read1.signal"))[:6000]
    with open(inp / "a.signal", "w") as f:
        f.write(" ".join(str(int(v)) for v in sig))
    with open(inp / "b.signal", "w") as f:
        f.write(" ".join(str(int(v)) for v in sig[1000:4000]))
    FLAGS = serve.ClientFlags(str(inp), str(tmp_path / "out"), "%s:%d" % srv.address, batch_size=7, concurrency=3, authkey=srv.authkey)
    assert FLAGS.segment_len == 400 and FLAGS.jump == 30
    res = serve.do_inference(FLAGS)
    assert set(res) == {"a", "b"}
    big = StandInEngine(max_batch=4096)
    for stem, path in (("a", inp / "a.signal"), ("b", inp / "b.signal")):
        ds = signal_io.read_data_for_eval(str(path), 0, 30, 400)
        r = big.infer(ds.event, seq_len_for_engine(ds.event_length, 4.0))
        (reads,), (uniq,) = ce.sparse2dense(([r.decoded], None))
        from chiron_amd import assembly
        bp = [ce.index2base(x) for x in reads]
        cons, cqs = assembly.simple_assembly_qs(bp, r.prob_logits[uniq], 30 / 400, kernal=ce.get_assembler_kernal(30, 400))
        want = ce.index2base(np.argmax(cons, axis=0))
        assert res[stem] == want
        fq = open(os.path.join(FLAGS.output, "result", stem + ".fastq")).read().split("\n")
        assert fq[0] == "@" + stem and fq[1] == want and len(fq[3]) == len(want)
        assert os.path.exists(os.path.join(FLAGS.output, "segments", stem + ".fastq"))
        assert os.path.exists(os.path.join(FLAGS.output, "meta", stem + ".meta"))
    with pytest.raises(ValueError):
        serve.ClientFlags(str(inp), str(tmp_path), "x:1", mode="protein")


def test_wire_frames_are_not_pickles_and_are_validated(served):
    """The wire carries a JSON header + raw little-endian tensors; nothing is unpickled.  A pickle payload, a tensor
    overrunning its frame, an unknown dtype and trailing bytes are all refused with an error reply, and the
    connection keeps working."""
    import pickle
    from multiprocessing.connection import Client
    eng, srv = served
    f, t = serve.unpack_frame(serve.pack_frame({"a": 1}, {"x": np.arange(6, dtype=np.int32).reshape(2, 3), "skip": None}))
    assert f == {"a": 1} and t["x"].tolist() == [[0, 1, 2], [3, 4, 5]] and "skip" not in t
    with pytest.raises(TypeError):
        serve.pack_frame({}, {"o": np.array([object()])})
    for bad in (b"", b"\x05\x00\x00\x00{}", serve.pack_frame({}) + b"x",
                serve.pack_frame({}, {"x": np.zeros(4, np.float32)})[:-3]):
        with pytest.raises(ValueError):
            serve.unpack_frame(bad)
    c = Client(srv.address, authkey=srv.authkey)
    try:
        for payload in (pickle.dumps({"method": "signature"}), b"\x00" * 3,
                        serve.pack_frame({"method": "predict"}, {"x": np.zeros((1, 400), np.float32)})):
            c.send_bytes(payload)
            rep, _ = serve.unpack_frame(c.recv_bytes())
            assert "error" in rep
        c.send_bytes(serve.pack_frame({"method": "signature"}))
        assert serve.unpack_frame(c.recv_bytes())[0]["T"] == 100
    finally:
        c.close()
    with pytest.raises(ValueError):
        serve.PredictClient(srv.address, None)
    assert len(serve.make_authkey()) == 48 and serve.make_authkey() != serve.make_authkey()


def test_server_main_reaches_engine_creation(tmp_path, monkeypatch):
    """`python -m chiron_amd.serve server ...` up to the engine: load_model's (spec, weights, config) is unpacked, the
    key file is created with mode 0600, and the library's refusal to run without a GPU surfaces as its own error."""
    import argparse
    from chiron_amd import _lib
    key = tmp_path / "key"
    import json
    model = tmp_path / "model"                       # no checkpoint: the index-less default topology + synthetic weights
    model.mkdir()
    (model / "model.json").write_text(json.dumps({"cnn": {"model": "dna_model1"}, "rnn": {"layer_num": 3, "hidden_num": 100,
                                                                                     "cell_type": "LSTM", "layer_type": "normal"}}))
    a = argparse.Namespace(model=str(model), port=0, mode="dna", batch_size=4, segment_len=None,
                           beam=0, slots=1, synthetic_weights=True, authkey_file=str(key))
    try:
        srv, eng = serve.start_server(a)
    except _lib.ChironError as exc:                  # CPU box: engine creation is where it must stop
        assert exc.status == _lib.ERR_DEVICE
    else:
        srv.close()
        eng.close()
    assert key.exists() and (key.stat().st_mode & 0o777) == 0o600 and len(key.read_bytes()) == 48
