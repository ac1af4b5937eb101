"""Python face of the C ABI: one Engine per GPU.

Mirrors the reference seam `chiron_model.inference` (chiron_model.py:134-172) +
the decode sub-graph (chiron_eval.py:465-492): feed (x[B,L] f32, seq_len[B]
i32) -> (SparseTensor(indices, values, dense_shape), log_prob, prob_logits,
logits), the SavedModel PREDICT signature of export_test.py:103-112.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from .model import ModelSpec

SparseTensor = namedtuple("SparseTensor", "indices values dense_shape")      # chiron_eval.py:34
# compact: None, or CompactDecode -- the decode in the per-row form of the regroup step (CHIRON_COMPACT_DECODE): `flat` uint8 [nnz], the
# rows' labels back to back in row order; `counts` int32 [batch], labels per row.  Then `decoded` is None: the (indices, values) pair
# of the SparseTensor is the same information spread over 24 bytes per base.
CompactDecode = namedtuple("CompactDecode", "flat counts dense_shape")
DecodeResult = namedtuple("DecodeResult", "decoded log_prob prob_logits logits compact", defaults=(None,))


def _is_torch_cuda(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and x.is_cuda


_DTYPES = {"fp32": _lib.F32, "fp16": _lib.F16, "fp32-split": _lib.F32_SPLIT, "fp16-w2": _lib.F16_W2}


def plan_sizes(spec, max_batch, segment_len, n_slots=1, dtype="fp32", max_beam=0):
    """chiron_engine_plan: what an Engine of this shape would allocate, without a GPU.  Raises ChironError with status
    ERR_OVERFLOW when a tensor would exceed what the kernels can address (the engine constructor refuses the same)."""
    desc = spec.to_c()
    opts = _lib.EngineOpts(0, max_batch, segment_len, n_slots, _DTYPES[dtype], max_beam)
    out = _lib.EngineSizes()
    _lib.check(_lib.load().chiron_engine_plan(C.byref(desc), C.byref(opts), C.byref(out)))
    return {k: getattr(out, k) for k, _ in _lib.EngineSizes._fields_}


class Engine(object):
    def __init__(self, spec, weights, max_batch, segment_len, device_id=0, n_slots=1, max_beam=0, dtype="fp32", calibrate=False):
        """calibrate (dtype "fp16" only; opt-in, a no-op for every other dtype): apply the bias correction for the weights' rounding to
        halves on the fixed synthetic calibration batch (calibration_windows) right after creation.  THE one place every entry point
        -- `chiron call` (eval.py), the serving surface (serve.py), bench.py, direct users -- asks for it, so that the same read
        decodes to the same string whichever of them built the engine; `self.calibrated` says whether it was applied and `chiron
        call` records it in <output>/log/engine*.json.  The correction was validated on synthetic trained-like weights and synthetic
        signal: with a real checkpoint compare both settings on real reads before relying on it."""
        if not isinstance(spec, ModelSpec):
            raise TypeError("spec must be a ModelSpec")
        self._lib = _lib.load()
        self.spec = spec
        blob = spec.pack(weights) if isinstance(weights, dict) else np.ascontiguousarray(weights, dtype=np.float32)
        desc = spec.to_c()
        need = C.c_size_t()
        _lib.check(self._lib.chiron_weights_size(C.byref(desc), C.byref(need)))
        if need.value != blob.size:
            raise ValueError("weight blob has %d floats, descriptor needs %d" % (blob.size, need.value))
        opts = _lib.EngineOpts(device_id, max_batch, segment_len, n_slots,
                               _DTYPES[dtype], max_beam)
        h = C.c_void_p()
        _lib.check(self._lib.chiron_engine_create(C.byref(desc), blob.ctypes.data_as(C.c_void_p), blob.size,
                                                  C.byref(opts), C.byref(h)))
        self._h = h
        t, r = C.c_int32(), C.c_double()
        _lib.check(self._lib.chiron_engine_dims(self._h, C.byref(t), C.byref(r)))
        self.T = t.value
        self.ratio = r.value
        self.max_batch = max_batch
        self.segment_len = segment_len
        self.n_slots = n_slots
        self.device_id = device_id
        self._keep = [None] * n_slots      # keep submitted arrays alive until collect
        self.dtype = dtype
        self.calibrated = False
        if calibrate and dtype == "fp16":
            self.calibrate()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.chiron_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------
    def submit(self, slot, x, seq_len, beam_width=0, want_prob=True, want_logits=False, copy_decoded=True, compact=False):
        """Asynchronous: enqueue one batch on `slot`'s stream.  x/seq_len may be
        numpy arrays (host) or torch CUDA tensors already resident on this
        engine's device (zero-copy)."""
        flags = 0
        if _is_torch_cuda(x):
            if not (_is_torch_cuda(seq_len)):
                raise TypeError("x on device requires seq_len on device too")
            import torch
            if x.dtype != torch.float32 or seq_len.dtype != torch.int32:
                raise TypeError("device inputs must be float32 / int32")
            x = x.contiguous()
            seq_len = seq_len.contiguous()
            batch = x.shape[0]
            if x.dim() != 2 or x.shape[1] != self.segment_len:
                raise ValueError("x must be [batch, %d]" % self.segment_len)
            xp, sp = C.c_void_p(x.data_ptr()), C.c_void_p(seq_len.data_ptr())
            flags |= _lib.X_ON_DEVICE
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
            if x.ndim != 2 or x.shape[1] != self.segment_len:
                raise ValueError("x must be [batch, %d], got %s" % (self.segment_len, x.shape))
            batch = x.shape[0]
            xp, sp = x.ctypes.data_as(C.c_void_p), seq_len.ctypes.data_as(C.c_void_p)
        if seq_len.shape[0] != batch:
            raise ValueError("seq_len must have one entry per row")
        if want_prob:
            flags |= _lib.WANT_PROB
        if want_logits:
            flags |= _lib.WANT_LOGITS
        if not copy_decoded:
            flags |= _lib.NO_DECODE_COPY
        if compact:
            flags |= _lib.COMPACT_DECODE
        # A submit refused with ERR_STATE leaves the slot's in-flight batch untouched: its keep-alive must survive
        _lib.check(self._lib.chiron_engine_submit(self._h, slot, xp, sp, batch, int(beam_width), flags))
        self._keep[slot] = (x, seq_len)

    def submit_pieces(self, slot, pieces, seq_len, beam_width=0, want_prob=True, compact=True):
        """submit() with the batch given as a list of C-contiguous float32 arrays of whole rows [n_i, segment_len] (the tail of one
        read, whole reads, the head of the next: chiron_eval.py:321-334): chiron_engine_submit_pieces copies them straight into the
        slot's pinned staging buffer, the [batch, segment_len] array is never built.  compact: ask for the decode in the regroup's
        per-row form (collect().compact) instead of the SparseTensor."""
        n = len(pieces)
        ptrs = (C.c_void_p * n)()
        rows = (C.c_int32 * n)()
        strides = (C.c_int64 * n)()
        for i, a in enumerate(pieces):
            if not piece_ok(a, self.segment_len):
                raise ValueError("piece %d must be a float32 [n, %d] array with contiguous rows (row stride any positive multiple of 4 bytes: "
                                 "windows of one signal buffer are welcome)" % (i, self.segment_len))
            ptrs[i] = a.ctypes.data
            rows[i] = a.shape[0]
            strides[i] = a.strides[0] // 4 if a.shape[0] > 1 else self.segment_len
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        flags = (_lib.WANT_PROB if want_prob else 0) | (_lib.COMPACT_DECODE if compact else 0)
        _lib.check(self._lib.chiron_engine_submit_pieces(self._h, slot, ptrs, rows, strides, n, seq_len.ctypes.data_as(C.c_void_p),
                                                         int(seq_len.shape[0]), int(beam_width), flags))
        self._keep[slot] = (pieces, seq_len)

    def collect(self, slot):
        """Blocks; returns DecodeResult with numpy copies (valid indefinitely)."""
        d = _lib.Decoded()
        _lib.check(self._lib.chiron_engine_collect(self._h, slot, C.byref(d)))
        self._keep[slot] = None
        nnz, B, T, K = d.nnz, d.batch, d.T, self.spec.classes
        if d.row_counts:       # CHIRON_COMPACT_DECODE: the SparseTensor was not copied
            flat = np.ctypeslib.as_array(d.flat_labels, shape=(nnz,)).copy() if nnz > 0 else np.zeros(0, dtype=np.uint8)
            comp = CompactDecode(flat, np.ctypeslib.as_array(d.row_counts, shape=(B,)).copy(),
                                 np.asarray([d.dense_shape[0], d.dense_shape[1]], dtype=np.int64))
            lp = np.ctypeslib.as_array(d.log_prob, shape=(B, 1)).copy()
            pr = np.ctypeslib.as_array(d.prob_logits, shape=(B, 1)).copy()
            return DecodeResult(None, lp, pr, None, comp)
        if nnz > 0 and d.indices:
            idx = np.ctypeslib.as_array(d.indices, shape=(nnz, 2)).copy()
            val = np.ctypeslib.as_array(d.values, shape=(nnz,)).copy()
        else:
            idx = np.zeros((0, 2), dtype=np.int64)
            val = np.zeros((0,), dtype=np.int64)
        shape = np.asarray([d.dense_shape[0], d.dense_shape[1]], dtype=np.int64)
        lp = np.ctypeslib.as_array(d.log_prob, shape=(B, 1)).copy()
        pr = np.ctypeslib.as_array(d.prob_logits, shape=(B, 1)).copy()
        lg = np.ctypeslib.as_array(d.logits, shape=(B, T, K)).copy() if d.logits else None
        return DecodeResult(SparseTensor(idx, val, shape), lp, pr, lg)

    def infer(self, x, seq_len, beam_width=0, want_prob=True, want_logits=False, slot=0):
        self.submit(slot, x, seq_len, beam_width, want_prob, want_logits)
        return self.collect(slot)

    def decode(self, logits, seq_len, beam_width=0, want_prob=True, slot=0):
        """Decode-only (chiron_eval.decoding_queue): logits float32 [batch, T, K] on the host."""
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        if logits.ndim != 3 or logits.shape[1] != self.T or logits.shape[2] != self.spec.classes:
            raise ValueError("logits must be [batch, %d, %d]" % (self.T, self.spec.classes))
        flags = _lib.WANT_PROB if want_prob else 0
        _lib.check(self._lib.chiron_engine_decode(self._h, slot, logits.ctypes.data_as(C.c_void_p),
                                                  seq_len.ctypes.data_as(C.c_void_p), logits.shape[0],
                                                  int(beam_width), flags))
        self._keep[slot] = (logits, seq_len)
        return self.collect(slot)

    def features(self, slot=0):
        """getcnnfeature (cnn.py:334-371): the CNN feature tensor [batch, T, C] of the batch last run on the (idle) slot."""
        b, c = C.c_int32(), C.c_int32()
        st = self._lib.chiron_engine_features(self._h, slot, None, 0, C.byref(b), C.byref(c))
        if st != _lib.ERR_OVERFLOW:
            _lib.check(st)
        out = np.empty((b.value, self.T, c.value), dtype=np.float32)
        _lib.check(self._lib.chiron_engine_features(self._h, slot, out.ctypes.data_as(C.c_void_p), out.size, C.byref(b), C.byref(c)))
        return out

    def calibrate(self, x=None, seq_len=None, iterations=2):
        """f16 engines: bias correction for the weights' rounding to halves (chiron_engine_calibrate); a no-op for fp32 / fp32-split.
        x [n, segment_len] float32 calibration windows (default: `calibration_windows`, a fixed synthetic squiggle, so that every
        process that builds this engine builds the same one); iterations = 0 restores the uncorrected engine."""
        if int(iterations) == 0:
            _lib.check(self._lib.chiron_engine_calibrate(self._h, None, None, 0, 0))
            self.calibrated = False
            return
        if x is None:
            x, seq_len = calibration_windows(self.segment_len, min(self.max_batch, 256), self.ratio)
        x = np.ascontiguousarray(x, dtype=np.float32)
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        if x.ndim != 2 or x.shape[1] != self.segment_len or seq_len.shape[0] != x.shape[0]:
            raise ValueError("x must be [n, %d] with one seq_len per row" % self.segment_len)
        _lib.check(self._lib.chiron_engine_calibrate(self._h, x.ctypes.data_as(C.c_void_p), seq_len.ctypes.data_as(C.c_void_p),
                                                     x.shape[0], int(iterations)))
        self.calibrated = self.dtype == "fp16"

    def rnn_output(self, slot=0):
        """`lasth` (rnn.py:63-65 / :140-145): the recurrent stack's output [batch, T, 2H] of the batch last run on the (idle) slot."""
        b, w = C.c_int32(), C.c_int32()
        st = self._lib.chiron_engine_rnn_output(self._h, slot, None, 0, C.byref(b), C.byref(w))
        if st != _lib.ERR_OVERFLOW:
            _lib.check(st)
        out = np.empty((b.value, self.T, w.value), dtype=np.float32)
        _lib.check(self._lib.chiron_engine_rnn_output(self._h, slot, out.ctypes.data_as(C.c_void_p), out.size, C.byref(b), C.byref(w)))
        return out

    def sync(self):
        _lib.check(self._lib.chiron_engine_sync(self._h))

    def device_results(self, slot):
        p = [C.c_void_p() for _ in range(4)]
        _lib.check(self._lib.chiron_engine_device_results(self._h, slot, *[C.byref(q) for q in p]))
        return tuple(q.value for q in p)

    # ------------------------------------------------------------------
    def profile(self, enable=True):
        _lib.check(self._lib.chiron_engine_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        arr = (_lib.KernelStat * 32)()
        n = C.c_int32()
        _lib.check(self._lib.chiron_engine_profile_read(self._h, arr, 32, C.byref(n)))
        out = {}
        for i in range(n.value):
            s = arr[i]
            out[s.name.decode()] = {"total_ms": s.total_ms, "launches": s.launches, "flops": s.flops, "bytes": s.bytes}
        return out


def piece_ok(a, segment_len):
    """what Engine.submit_pieces takes as one piece: float32 [n, segment_len], every row contiguous, rows any positive whole number of
    floats apart (C-contiguous arrays, row slices of them, and the overlapping windows signal_io.window_signal returns)"""
    return (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == segment_len and a.strides[1] == 4
            and (a.shape[0] <= 1 or (a.strides[0] > 0 and a.strides[0] % 4 == 0)))


def calibration_windows(segment_len, n, ratio, seed=20260928):
    """The fixed calibration batch of `chiron call --dtype fp16`: n full windows of a seeded synthetic 4 kHz squiggle (model.
    synthetic_signal: dwell ~ 8.9 samples per level, levels N(500, 80) clipped to [200, 1000]) -> (x, seq_len)."""
    from .model import synthetic_signal
    sig = synthetic_signal(1, segment_len * n, seed=seed)[0]
    x = np.ascontiguousarray(sig[:segment_len * n].reshape(n, segment_len), dtype=np.float32)
    return x, seq_len_for_engine(np.full(n, segment_len), ratio)


def seq_len_for_engine(lengths, ratio):
    """chiron_eval.py:337: np.round(seq_len/ratio).astype(np.int32) (round half even)."""
    return np.round(np.asarray(lengths, dtype=np.float64) / ratio).astype(np.int32)
