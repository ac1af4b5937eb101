"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/chiron_oracle.c (fp32 C
restatement; checker for larger sizes and the timed CPU baseline of bench.py)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libchiron_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("%s missing: run `make -C oracle` (or __graft_entry__.build())" % _PATH)
        _lib = C.CDLL(_PATH)
        _lib.chiron_oracle_forward.restype = C.c_int
        _lib.chiron_oracle_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p, C.c_int]
        _lib.chiron_oracle_greedy.restype = None
        _lib.chiron_oracle_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.chiron_oracle_max_threads.restype = C.c_int
        _lib.chiron_oracle_beam.restype = C.c_int
        _lib.chiron_oracle_beam.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        for fn in ("chiron_oracle_ctc_exp", "chiron_oracle_ctc_log"):
            getattr(_lib, fn).restype = C.c_float
            getattr(_lib, fn).argtypes = [C.c_float]
        _lib.chiron_oracle_ctc_lse.restype = C.c_float
        _lib.chiron_oracle_ctc_lse.argtypes = [C.c_float, C.c_float]
    return _lib


def ctc_exp(d):
    """the decoder's e^d (d <= 0), see chiron_oracle.c exp_neg_"""
    return float(load().chiron_oracle_ctc_exp(float(d)))


def ctc_log(x):
    return float(load().chiron_oracle_ctc_log(float(x)))


def ctc_lse(a, b):
    return float(load().chiron_oracle_ctc_lse(float(a), float(b)))


def _desc(spec):
    if spec.get("stem"):
        raise NotImplementedError("the C oracle covers the shipped topologies; stem models are checked with nn_oracle")
    d = [len(spec["cnn"])]
    for b in spec["cnn"]:
        d += [b["in"], b["out"], b["k"], b.get("stride", 1), int(bool(b["i_bn"]))]
    d += [0 if spec["rnn"]["kind"] == "stack" else 1, spec["rnn"]["layers"], spec["rnn"]["hidden"], spec["classes"],
          0 if spec["bn_mode"] == "population" else 1]
    return np.asarray(d, dtype=np.int32)


def max_threads():
    return load().chiron_oracle_max_threads()


def forward(x, seq_len, spec, blob, T, threads=0):
    """x [B,L] f32, blob = weights in ABI order (include/chiron_amd.h) -> logits [B,T,K] f32."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    sl = np.ascontiguousarray(seq_len, dtype=np.int32)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    d = _desc(spec)
    B, L = x.shape
    out = np.empty((B, T, spec["classes"]), dtype=np.float32)
    t = lib.chiron_oracle_forward(d.ctypes.data, blob.ctypes.data, x.ctypes.data, sl.ctypes.data, B, L,
                                  out.ctypes.data, threads)
    if t != T:
        raise RuntimeError("C oracle returned T=%d, expected %d" % (t, T))
    return out


def greedy(logits, seq_len):
    lib = load()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    sl = np.ascontiguousarray(seq_len, dtype=np.int32)
    B, T, K = lg.shape
    labels = np.zeros((B, T), dtype=np.uint8)
    count = np.zeros(B, dtype=np.int32)
    neg = np.zeros(B, dtype=np.float32)
    pp = np.zeros(B, dtype=np.float32)
    lib.chiron_oracle_greedy(lg.ctypes.data, sl.ctypes.data, B, T, K, labels.ctypes.data, count.ctypes.data,
                             neg.ctypes.data, pp.ctypes.data)
    rows = [labels[b, :count[b]].astype(int).tolist() for b in range(B)]
    return rows, neg.reshape(-1, 1), pp.reshape(-1, 1)


def beam(logits, seq_len, beam_width):
    """float32 restatement of TF's sequential CTC beam search (top path) -> (rows, log_prob [B,1])."""
    lib = load()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    sl = np.ascontiguousarray(seq_len, dtype=np.int32)
    B, T, K = lg.shape
    labels = np.zeros((B, T), dtype=np.uint8)
    count = np.zeros(B, dtype=np.int32)
    lp = np.zeros(B, dtype=np.float32)
    rc = lib.chiron_oracle_beam(lg.ctypes.data, sl.ctypes.data, B, T, K, int(beam_width), labels.ctypes.data,
                                count.ctypes.data, lp.ctypes.data)
    if rc != 0:
        raise RuntimeError("chiron_oracle_beam failed: %d" % rc)
    rows = [labels[b, :count[b]].astype(int).tolist() for b in range(B)]
    return rows, lp.reshape(-1, 1)
