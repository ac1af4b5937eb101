"""Overlap-consensus assembly (counterpart of chiron/utils/easy_assembler.py).

glue / stick run in the native library (chiron_assemble, csrc/assemble.cpp); the 'simple' kernel
(jump <= 0.9*segment_len) needs difflib's Ratcliff-Obershelp matching blocks, which is Python
stdlib in the reference too, and stays in Python."""
import ctypes as C
import difflib
import math
from itertools import groupby

import numpy as np

from . import _lib

_BASE = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}


def mapping(full_path, blank_pos=4):
    """easy_assembler.py:26-34: merge repeats, drop blanks."""
    merged = np.asarray([k for k, _ in groupby(np.asarray(full_path))])
    return np.delete(merged, np.argwhere(merged == blank_pos))


def glue_kernal(bpread, prev_bpread):
    """easy_assembler.py:276-294 (pure-Python form; the batch path uses the native kernel)."""
    prev_n, n = len(prev_bpread), len(bpread)
    max_overlap = min(math.floor(0.1 * prev_n), n)
    best = (0, 0)
    for i in range(1, max_overlap):
        score = 2 * sum(a == b for a, b in zip(bpread[:i], prev_bpread[-i:])) - i
        if score > best[1]:
            best = (i, score)
    return prev_n - best[0]


def stick_kernal(bpread, prev_bpread):
    """easy_assembler.py:296-300."""
    return len(prev_bpread)


def simple_assembly_kernal(bpread, prev_bpread, error_rate, jump_step_ratio):
    """easy_assembler.py:212-250, constants reproduced literally."""
    back_ratio = 6.5 * 10e-4
    p_same = 1 - 2 * error_rate + 26 / 25 * (error_rate ** 2)
    p_diff = 1 - p_same
    ns, nd, log_px = dict(), dict(), dict()
    N = len(bpread)
    blocks = difflib.SequenceMatcher(a=bpread, b=prev_bpread).get_matching_blocks()
    for block in blocks:
        offset = block[1] - block[0]
        ns[offset] = ns.get(offset, 0) + block[2]
        nd[offset] = 0
    for key in ns.keys():
        if key < 0:
            k = -key
            log_px[key] = k * np.log((back_ratio) * N * jump_step_ratio) - sum([np.log(x + 1) for x in range(k)]) + \
                ns[key] * np.log(p_same / 0.25) + nd[key] * np.log(p_diff / 0.25)
        else:
            log_px[key] = key * np.log(N * jump_step_ratio) - sum([np.log(x + 1) for x in range(key)]) + \
                ns[key] * np.log(p_same / 0.25) + nd[key] * np.log(p_diff / 0.25)
    disp = max(log_px.keys(), key=lambda x: log_px[x])
    return disp, log_px[disp]


def add_count(concensus, start_indx, segment):
    """easy_assembler.py:381-387."""
    if start_indx < 0:
        segment = segment[-start_indx:]
        start_indx = 0
    for i, base in enumerate(segment):
        concensus[_BASE[base]][start_indx + i] += 1


def add_count_qs(concensus, concensus_qs, start_indx, segment, qs):
    """easy_assembler.py:435-442."""
    if start_indx < 0:
        segment = segment[-start_indx:]
        start_indx = 0
    for i, base in enumerate(segment):
        concensus[_BASE[base]][start_indx + i] += 1
        concensus_qs[_BASE[base]][start_indx + i] += qs[0]


def assemble_native(bases, seg_off, seg_qs, kernal):
    """chiron_assemble (include/chiron_amd.h): bases uint8 0..3 concatenated, seg_off int64
    [n+1].  -> (counts [4,len] float64, qs_sum [4,len] float64 or None)."""
    lib = _lib.load()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    n_seg = seg_off.shape[0] - 1
    kid = {"glue": _lib.KERNAL_GLUE, "stick": _lib.KERNAL_STICK}[kernal]
    qs = None if seg_qs is None else np.ascontiguousarray(seg_qs, dtype=np.float64).ravel()
    cap = int(bases.shape[0]) + 1
    counts = np.empty((4, cap), dtype=np.float64)
    qsum = np.empty((4, cap), dtype=np.float64) if qs is not None else None
    out_len = C.c_int64()
    _lib.check(lib.chiron_assemble(bases.ctypes.data, seg_off.ctypes.data, n_seg,
                                   None if qs is None else qs.ctypes.data, kid, counts.ctypes.data,
                                   None if qsum is None else qsum.ctypes.data, cap, C.byref(out_len)))
    n = out_len.value
    return counts[:, :n].copy(), (None if qsum is None else qsum[:, :n].copy())


def _encode(bpreads):
    lens = np.fromiter((len(r) for r in bpreads), dtype=np.int64, count=len(bpreads))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.frombuffer("".join(bpreads).encode("ascii"), dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint8)
    for ch, v in _BASE.items():
        lut[ord(ch)] = v
    return lut[flat], off


def _python_assembly(bpreads, qs_list, jump_step_ratio, error_rate, kernal):
    """easy_assembler.py:302-335 / :393-432 structure for the Python-only kernels."""
    census_len = 1000
    concensus = np.zeros([4, census_len])
    concensus_qs = np.zeros([4, census_len]) if qs_list is not None else None
    pos = 0
    length = 0
    for indx, bpread in enumerate(bpreads):
        if indx == 0:
            if qs_list is None:
                add_count(concensus, 0, bpread)
            else:
                add_count_qs(concensus, concensus_qs, 0, bpread, qs_list[indx])
            continue
        prev_bpread = bpreads[indx - 1]
        if kernal == "simple":
            disp, _ = simple_assembly_kernal(bpread, prev_bpread, error_rate, jump_step_ratio)
        elif kernal == "glue":
            disp = glue_kernal(bpread, prev_bpread)
        elif kernal == "stick":
            disp = stick_kernal(bpread, prev_bpread)
        else:
            raise ValueError("assembly kernal %r is not available (the reference hard-codes 'simple', "
                             "chiron_eval.py:144-145; 'global' needs Bio.pairwise2)" % kernal)
        while disp + pos + len(bpread) > census_len:
            concensus = np.pad(concensus, ((0, 0), (0, 1000)), mode="constant", constant_values=0)
            if concensus_qs is not None:
                concensus_qs = np.pad(concensus_qs, ((0, 0), (0, 1000)), mode="constant", constant_values=0)
            census_len += 1000
        if qs_list is None:
            add_count(concensus, pos + disp, bpread)
        else:
            add_count_qs(concensus, concensus_qs, pos + disp, bpread, qs_list[indx])
        pos += disp
        length = max(length, pos + len(bpread))
    if qs_list is None:
        return concensus[:, :length]
    return concensus[:, :length], concensus_qs[:, :length]


def simple_assembly(bpreads, jump_step_ratio, error_rate=0.2, kernal="global"):
    """easy_assembler.py:302-335."""
    if kernal in ("glue", "stick") and len(bpreads) > 0:
        bases, off = _encode(bpreads)
        return assemble_native(bases, off, None, kernal)[0]
    return _python_assembly(bpreads, None, jump_step_ratio, error_rate, kernal)


def simple_assembly_qs(bpreads, qs_list, jump_step_ratio, error_rate=0.2, kernal="global"):
    """easy_assembler.py:393-432."""
    assert len(bpreads) == len(qs_list)
    if kernal in ("glue", "stick") and len(bpreads) > 0:
        bases, off = _encode(bpreads)
        return assemble_native(bases, off, np.asarray(qs_list, dtype=np.float64).reshape(len(bpreads), -1)[:, 0], kernal)
    return _python_assembly(bpreads, qs_list, jump_step_ratio, error_rate, kernal)
