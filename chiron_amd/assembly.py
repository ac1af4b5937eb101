"""Overlap-consensus assembly: the Python face of chiron_assemble / chiron_overlap_displacement (csrc/assemble.cpp).

Counterpart of chiron/utils/easy_assembler.py for the kernels `chiron call` can select (chiron_eval.py:138-150):
glue (jump > 0.9 L), stick (jump >= L) and simple (everything else: difflib matching blocks + offset prior).  All
three run in the native library; nothing here computes a displacement or a vote in Python.  Function names and
argument order follow the reference so call sites read the same.
"""
import ctypes as C

import numpy as np

from . import _lib

KERNALS = {"glue": _lib.KERNAL_GLUE, "stick": _lib.KERNAL_STICK, "simple": _lib.KERNAL_SIMPLE}
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _CODE[ord(_ch)] = _CODE[ord(_ch.lower())] = _i


def _kernal_id(kernal):
    if kernal not in KERNALS:
        raise ValueError("assembly kernal %r is not available (the reference hard-codes 'simple', "
                         "chiron_eval.py:144-145; 'global' needs Bio.pairwise2)" % (kernal,))
    return KERNALS[kernal]


def mapping(full_path, blank_pos=4):
    """easy_assembler.py:26-34 -- collapse runs, then drop blanks."""
    path = np.asarray(full_path)
    if path.size == 0:
        return path
    first_of_run = np.ones(path.shape[0], dtype=bool)
    first_of_run[1:] = path[1:] != path[:-1]
    return path[first_of_run & (path != blank_pos)]


def encode(bpreads):
    """list of base strings -> (codes uint8 0..3 concatenated, offsets int64 [n+1])"""
    off = np.zeros(len(bpreads) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in bpreads], out=off[1:])
    codes = _CODE[np.frombuffer("".join(bpreads).encode("ascii"), dtype=np.uint8)]
    if codes.size and codes.max() > 3:
        raise ValueError("segments may only hold the bases A, C, G, T")
    return codes, off


def displacement(bpread, prev_bpread, kernal, error_rate=0.2, jump_step_ratio=1.0):
    """Where `bpread` starts relative to the start of `prev_bpread` -> (disp, log_px) (log_px is 0 except for 'simple')."""
    cur, _ = encode([bpread])
    prev, _ = encode([prev_bpread])
    disp, log_px = C.c_int64(), C.c_double()
    _lib.check(_lib.load().chiron_overlap_displacement(cur.ctypes.data, cur.size, prev.ctypes.data, prev.size, _kernal_id(kernal),
                                                       float(error_rate), float(jump_step_ratio), C.byref(disp), C.byref(log_px)))
    return disp.value, log_px.value


def glue_kernal(bpread, prev_bpread):
    """easy_assembler.py:276-294."""
    return displacement(bpread, prev_bpread, "glue")[0]


def stick_kernal(bpread, prev_bpread):
    """easy_assembler.py:296-300."""
    return displacement(bpread, prev_bpread, "stick")[0]


def simple_assembly_kernal(bpread, prev_bpread, error_rate, jump_step_ratio):
    """easy_assembler.py:212-250 -> (disp, log_px[disp])."""
    return displacement(bpread, prev_bpread, "simple", error_rate, jump_step_ratio)


def assemble_native(bases, seg_off, seg_qs, kernal, error_rate=0.2, jump_step_ratio=1.0):
    """chiron_assemble on pre-encoded segments -> (counts [4, len] float64, qs_sum [4, len] float64 or None)."""
    lib = _lib.load()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    qs = None if seg_qs is None else np.ascontiguousarray(seg_qs, dtype=np.float64).ravel()
    kid = _kernal_id(kernal)
    # glue / stick never place a segment left of its predecessor's start, so the total base count bounds the length;
    # the simple kernel can, in principle, step further right than a segment is long: ask for the size when it does
    cap = int(bases.shape[0]) + 1
    for _ in range(2):
        counts = np.empty((4, cap), dtype=np.float64)
        qsum = np.empty((4, cap), dtype=np.float64) if qs is not None else None
        out_len = C.c_int64()
        st = lib.chiron_assemble(bases.ctypes.data, seg_off.ctypes.data, seg_off.shape[0] - 1, None if qs is None else qs.ctypes.data,
                                 kid, float(error_rate), float(jump_step_ratio), counts.ctypes.data,
                                 None if qsum is None else qsum.ctypes.data, cap, C.byref(out_len))
        if st == _lib.ERR_OVERFLOW and out_len.value > cap:
            cap = out_len.value
            continue
        _lib.check(st)
        break
    n = out_len.value
    return counts[:, :n].copy(), (None if qsum is None else qsum[:, :n].copy())


def simple_assembly(bpreads, jump_step_ratio, error_rate=0.2, kernal="global"):
    """easy_assembler.py:302-335 -> vote matrix [4, consensus length]."""
    _kernal_id(kernal)
    bases, off = encode(bpreads)
    return assemble_native(bases, off, None, kernal, error_rate, jump_step_ratio)[0]


def simple_assembly_qs(bpreads, qs_list, jump_step_ratio, error_rate=0.2, kernal="global"):
    """easy_assembler.py:393-432 -> (vote matrix, per-base sum of the voting segments' qualities)."""
    if len(bpreads) != len(qs_list):
        raise AssertionError("one quality per segment is required")
    _kernal_id(kernal)
    bases, off = encode(bpreads)
    seg_qs = np.asarray(qs_list, dtype=np.float64).reshape(len(bpreads), -1)[:, 0] if len(bpreads) else np.zeros(0)
    return assemble_native(bases, off, seg_qs, kernal, error_rate, jump_step_ratio)


def consensus_device(bpreads, qs_list, kernal, device_id=0):
    """chiron_consensus_device: glue / stick displacements, vote and argmax of ONE read on the GPU; the device returns the
    vote summary (n1, n2, quality sum behind the winner) and the Phred characters are computed here by eval.qs's own
    formula -> (consensus 'ACGT' string, Phred+33 string or None).  Equal to simple_assembly_qs + argmax + eval.qs."""
    kid = _kernal_id(kernal)
    if kernal == "simple":
        raise ValueError("the device vote covers the glue and stick kernels; simple displacements are host code")
    bases, off = encode(bpreads)
    qs = None if qs_list is None else np.ascontiguousarray(np.asarray(qs_list, dtype=np.float64).reshape(len(bpreads), -1)[:, 0])
    cap = int(bases.shape[0]) + 1
    cons = np.empty(cap, dtype=np.uint8)
    n1 = n2 = q_top = None
    if qs is not None:
        n1, n2, q_top = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.float64)
    n = C.c_int64()
    ptr = lambda a_: None if a_ is None else a_.ctypes.data
    _lib.check(_lib.load().chiron_consensus_device(int(device_id), bases.ctypes.data, off.ctypes.data, len(bpreads), ptr(qs), kid,
                                                   cons.ctypes.data, ptr(n1), ptr(n2), ptr(q_top), cap, C.byref(n)))
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[cons[:n.value]].tobytes().decode("ascii")
    if qs is None:
        return seq, None
    from .eval import qs_from_votes
    return seq, qs_from_votes(n1[:n.value].astype(np.float64), n2[:n.value].astype(np.float64), q_top[:n.value])
