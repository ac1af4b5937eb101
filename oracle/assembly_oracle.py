"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- Python restatement of the reference's overlap-consensus
assembly, the checker for the native chiron_assemble / chiron_overlap_displacement (chiron_amd/csrc/assemble.cpp).

Follows chiron/utils/easy_assembler.py: glue_kernal :276-294, stick_kernal :296-300, simple_assembly_kernal :212-250,
simple_assembly(_qs) :302-335 / :393-432 with add_count(_qs) :381-387 / :435-442.  The `simple` kernel rests on
difflib.SequenceMatcher, which is Python stdlib -- the same code the reference runs -- so this side uses it directly and
the native Ratcliff-Obershelp implementation is checked against the real thing.

PINNED: tests/test_host_golden.py holds both this file and the native code to vectors captured from the reference's own
functions (tests/golden/make_golden.py: 61 displacement pairs for glue / stick / simple at two jump ratios, assembled
vote matrices, and the reference's checked-in example reads).
"""
import difflib
import math

import numpy as np

BACK_RATIO = 6.5 * 10e-4       # easy_assembler.py:221, as written there (= 6.5e-3)


def glue_displacement(cur, prev):
    """Longest-suffix/prefix agreement within the last tenth of `prev`: score(i) = 2 * matches - i over overlaps
    i = 1 .. min(floor(0.1 * len(prev)), len(cur)) - 1, first strict maximum above 0 wins (easy_assembler.py:276-294)."""
    limit = min(math.floor(0.1 * len(prev)), len(cur))
    best_i, best_score = 0, 0
    for i in range(1, limit):
        same = sum(1 for x, y in zip(cur[:i], prev[len(prev) - i:]) if x == y)
        if 2 * same - i > best_score:
            best_i, best_score = i, 2 * same - i
    return len(prev) - best_i


def simple_displacement(cur, prev, error_rate, jump_step_ratio):
    """easy_assembler.py:212-250 -> (offset, score).  Offsets are diagonals (index in prev - index in cur) of difflib's
    matching blocks, the terminating empty block included; score = |off| ln(rate) - ln |off|! + matched ln(p_same / .25)
    (+ 0 * ln(p_diff / .25), kept because the reference adds it); first maximum in order of first appearance."""
    p_same = 1 - 2 * error_rate + 26 / 25 * (error_rate ** 2)
    gain_same, gain_diff = np.log(p_same / 0.25), np.log((1 - p_same) / 0.25)
    matched = {}
    for i, j, size in difflib.SequenceMatcher(a=cur, b=prev).get_matching_blocks():
        matched[j - i] = matched.get(j - i, 0) + size
    n = len(cur)
    best = None
    for off, same in matched.items():
        steps = abs(off)
        rate = (BACK_RATIO * n * jump_step_ratio) if off < 0 else (n * jump_step_ratio)
        score = steps * np.log(rate) - sum([np.log(x + 1) for x in range(steps)]) + same * gain_same + 0 * gain_diff
        if best is None or score > best[1]:
            best = (off, score)
    return best


def displacement(cur, prev, kernal, error_rate=0.2, jump_step_ratio=1.0):
    if kernal == "glue":
        return glue_displacement(cur, prev)
    if kernal == "stick":
        return len(prev)                      # easy_assembler.py:296-300
    if kernal == "simple":
        return simple_displacement(cur, prev, error_rate, jump_step_ratio)[0]
    raise ValueError(kernal)


def vote(segments, qualities, jump_step_ratio, error_rate, kernal):
    """easy_assembler.py:302-335 / :393-432: running start position, one vote per base, a segment that starts left of
    column 0 loses its head, and the consensus length counts segments 1.. only (segment 0 is `continue`d past the
    length update).  -> (counts [4, len], quality sums [4, len] or None), float64 like the reference's np.zeros."""
    index = {"A": 0, "C": 1, "G": 2, "T": 3}
    starts, pos, length = [0], 0, 0
    for k in range(1, len(segments)):
        pos += displacement(segments[k], segments[k - 1], kernal, error_rate, jump_step_ratio)
        starts.append(pos)
        length = max(length, pos + len(segments[k]))
    counts = np.zeros((4, length))
    qsum = np.zeros((4, length)) if qualities is not None else None
    for k, seg in enumerate(segments):
        for j, base in enumerate(seg):
            col = starts[k] + j
            if 0 <= col < length:
                counts[index[base.upper()], col] += 1
                if qsum is not None:
                    qsum[index[base.upper()], col] += np.asarray(qualities[k]).ravel()[0]
    return counts, qsum
