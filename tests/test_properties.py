"""CPU: property tests (hypothesis) of the host stages and of the oracle's decoders -- the size-independent properties SURVEY.md §4
asks for next to the golden vectors: whatever the lengths, jumps, batch sizes and strings are,

  * windowing covers the signal exactly once per `jump` and pads with zeros (chiron_input.py:276-286, :681-692),
  * the feed side (cross-read batch packing, wrap padding, chiron_eval.py:321-360) followed by the drain side (per-file regroup,
    chiron_eval.py:403-446) returns every read's windows once, in order, whatever batch they travelled in,
  * the signal text writer and parser are inverse to each other (extract_sig_ref.py:122-123, chiron_input.py:527-539),
  * greedy CTC decoding is the reference's own mapping() of the argmax path (easy_assembler.py:26-34) and a fixed point of itself,
  * the beam search with a beam wider than the number of label sequences is the exhaustive enumeration, and never scores a
    labelling above its exact probability,
  * the native glue / stick displacement kernels obey their definitions (easy_assembler.py:276-300) and the vote conserves votes.
"""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import chiron_amd as ca
from chiron_amd import assembly, signal_io, eval as ce
from oracle import ctc_oracle

SET = settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@SET
@given(n=st.integers(0, 3000), seg=st.integers(1, 400), jump=st.integers(1, 450), start=st.integers(0, 50), seed=st.integers(0, 2 ** 16))
def test_windowing_covers_the_signal(n, seg, jump, start, seed):
    sig = np.random.RandomState(seed).randint(200, 1000, size=n).astype(np.float32)
    ev, ln = signal_io.window_signal(sig, start, jump, seg)
    body = sig[start:]
    m = body.shape[0]
    assert ev.shape == (len(range(0, m, jump)), seg) and ev.dtype == np.float32 and ln.dtype == np.int32
    for j, i in enumerate(range(0, m, jump)):
        want = body[i:i + seg]
        assert ln[j] == want.shape[0] >= 1
        assert np.array_equal(ev[j, :ln[j]], want) and not ev[j, ln[j]:].any()
    if jump <= seg and m:          # no gaps: the first `jump` samples of every window, concatenated, are the signal
        rebuilt = np.concatenate([ev[j, :min(jump, ln[j])] for j in range(ev.shape[0])])
        assert np.array_equal(rebuilt, body)


class _Res(object):
    def __init__(self, decoded, log_prob, prob_logits):
        self.decoded, self.log_prob, self.prob_logits, self.logits = decoded, log_prob, prob_logits, None


@SET
@given(windows=st.lists(st.integers(0, 23), min_size=1, max_size=9), batch=st.integers(1, 17), seed=st.integers(0, 2 ** 16))
def test_packing_then_regrouping_returns_every_read_once_in_order(windows, batch, seed):
    """every window is 'decoded' to a string that names its read and its index: after packing across reads, wrap padding of the last
    batch and regrouping, read r holds its windows 0 .. n_r - 1 in order, whatever the batch size"""
    rng = np.random.RandomState(seed)
    L = 8
    packer = ce.BatchPacker(batch, L, 1.0)
    coll = ce.ReadCollector()
    got = {}

    def drain(b):
        # the canned decode of row k = [read id, window index] as base indices (mod 4), plus a marker length; rows of padding decode too
        rows = []
        for k in range(batch):
            rows.append([int(v) % 4 for v in b.x[k, :2]] + [0] * (int(b.x[k, 2]) % 3))
        idx, val, shape = ctc_oracle.rows_to_sparse(rows, batch)
        res = _Res(ca.SparseTensor(idx, val, shape), np.zeros((batch, 1), np.float32), np.arange(batch, dtype=np.float32).reshape(-1, 1))
        for name, flat, seg_len, qs_list, meta in coll.add_batch(b, res, True):
            got[name] = (flat, seg_len, qs_list)

    expect = {}
    for r, n in enumerate(windows):
        name = "read%d.signal" % r
        ev = np.zeros((n, L), dtype=np.float32)
        ev[:, 0], ev[:, 1] = r, np.arange(n)
        ev[:, 2] = rng.randint(0, 3, size=n)
        coll.expect(name, n, (0.0, 0.0))
        expect[name] = ev
        if n == 0:
            coll.val.pop(name, None)
            continue
        for b in packer.add_read(name, ev, np.full(n, L, dtype=np.int32)):
            assert b.x.shape == (batch, L) and b.n_valid == batch
            drain(b)
    last = packer.flush()
    if last is not None:
        assert last.x.shape == (batch, L) and 0 < last.n_valid <= batch
        assert all(f == "" for f in last.fname[last.n_valid:]) and (last.index[last.n_valid:] == -1).all()
        drain(last)
    assert not coll.val                                   # nothing half-finished
    for name, ev in expect.items():
        if ev.shape[0] == 0:
            continue
        flat, seg_len, qs_list = got[name]
        assert seg_len.shape[0] == ev.shape[0] and qs_list.shape[0] == ev.shape[0]
        ends = np.cumsum(seg_len)
        for k in range(ev.shape[0]):
            w = flat[ends[k] - seg_len[k]:ends[k]]
            assert w[0] == int(ev[k, 0]) % 4 and w[1] == k % 4 and seg_len[k] == 2 + int(ev[k, 2]) % 3


@SET
@given(vals=st.lists(st.integers(-32768, 32767), min_size=0, max_size=300), delim=st.sampled_from(["\n", " ", "\t", "\r\n", "  "]))
def test_signal_text_writer_and_parser_are_inverse(tmp_path_factory, vals, delim):
    from chiron_amd import fast5
    p = str(tmp_path_factory.mktemp("sig") / "a.signal")
    sig = np.asarray(vals, dtype=np.float32)
    fast5.write_signal_text(p, sig, delim)
    assert open(p, "rb").read().decode() == delim.join(str(v) for v in vals)
    back = signal_io.read_signal(p)
    assert back.dtype == np.float32 and np.array_equal(back, sig)


def _mapping(path, blank=4):
    """easy_assembler.py:26-34 mapping(): collapse repeats, drop the blank"""
    out, prev = [], -1
    for k in path:
        if k != prev and k != blank:
            out.append(k)
        prev = k
    return out


@SET
@given(data=st.data(), T=st.integers(1, 40), B=st.integers(1, 5))
def test_greedy_decode_is_mapping_of_the_argmax_path_and_a_fixed_point(data, T, B):
    logits = np.asarray(data.draw(st.lists(st.lists(st.lists(st.integers(-6, 6), min_size=5, max_size=5), min_size=T, max_size=T),
                                           min_size=B, max_size=B)), dtype=np.float32)       # integer logits: ties everywhere
    sl = np.asarray(data.draw(st.lists(st.integers(0, T), min_size=B, max_size=B)), dtype=np.int32)
    rows, nsl = ctc_oracle.greedy_decode(logits, sl)
    for b in range(B):
        path = [int(np.argmax(logits[b, t])) for t in range(sl[b])]                    # first maximum on ties
        assert rows[b] == _mapping(path)
        assert np.isclose(nsl[b, 0], -sum(float(logits[b, t, k]) for t, k in enumerate(path)))
        # fixed point: the one-hot path "label, blank, label, blank ..." of the decoded row decodes to the row
        hot = np.full((1, max(2 * len(rows[b]), 1), 5), -5.0, dtype=np.float32)
        hot[0, :, 4] = 0.0
        for j, k in enumerate(rows[b]):
            hot[0, 2 * j, k] = 5.0
        again, _ = ctc_oracle.greedy_decode(hot, np.asarray([hot.shape[1]]))
        assert again[0] == rows[b]
    idx, val, shape = ctc_oracle.rows_to_sparse(rows, B)
    assert shape[0] == B and shape[1] == max([len(r) for r in rows] + [0]) and len(val) == sum(len(r) for r in rows)
    assert all(tuple(idx[i]) < tuple(idx[i + 1]) for i in range(len(idx) - 1))         # row-major order


@settings(max_examples=25, deadline=None)
@given(data=st.data(), T=st.integers(1, 5))
def test_wide_beam_is_the_exhaustive_search(data, T):
    """a beam that can hold every prefix: the top path is the most probable LABELLING (sum over its alignments), and its score is that
    labelling's exact log-probability (oracle/ctc_oracle.brute_force_best enumerates all 5^T alignments)"""
    raw = np.asarray(data.draw(st.lists(st.lists(st.floats(-3, 3, allow_nan=False, width=32), min_size=5, max_size=5), min_size=T, max_size=T)),
                     dtype=np.float64)
    raw += np.arange(5)[None, :] * 1e-3 + np.arange(T)[:, None] * 1e-4          # no exact ties between labellings
    logits = raw[None].astype(np.float32)
    rows, lp = ctc_oracle.beam_search_decode(logits, np.asarray([T]), beam_width=4 ** T + 8)
    best, best_lp, _ = ctc_oracle.brute_force_best(logits[0], T)
    assert list(rows[0]) == list(best)
    assert abs(float(lp[0, 0]) - best_lp) < 1e-4


_bases = st.text(alphabet="ACGT", min_size=1, max_size=60)


@SET
@given(prev=_bases, cur=_bases)
def test_glue_and_stick_displacements_follow_their_definitions(built, prev, cur):
    """glue_kernal (easy_assembler.py:276-294): the FIRST strict maximum of 2 * matches(cur[:i], prev[-i:]) - i over i in 1 ..
    min(floor(0.1 * len(prev)), len(cur)) - 1, accepted only above 0, else pure append; stick_kernal (:296-300): always append."""
    d_glue = assembly.glue_kernal(cur, prev)
    d_stick = assembly.stick_kernal(cur, prev)
    assert d_stick == len(prev)
    best_i, best = 0, 0
    for i in range(1, min(int(np.floor(0.1 * len(prev))), len(cur))):
        score = 2 * sum(a == b for a, b in zip(cur[:i], prev[-i:])) - i
        if score > best:
            best_i, best = i, score
    assert d_glue == len(prev) - best_i


@SET
@given(segs=st.lists(_bases, min_size=0, max_size=8), kernal=st.sampled_from(["glue", "stick"]))
def test_the_vote_places_every_segment_where_its_displacements_say(built, segs, kernal):
    """simple_assembly (easy_assembler.py:302-335): segment k starts at the running sum of the displacements; the result is cut at
    `length`, which the reference only updates from the SECOND segment on -- so a read of one window has an empty consensus (kept:
    the reference's behaviour), and columns of the first segment past every later segment's end are cut off.  Within the kept columns
    every base votes exactly once: column sums = number of segments covering the column."""
    jr = 0.975 if kernal == "glue" else 1.0
    cons = assembly.simple_assembly(segs, jr, kernal=kernal)
    assert cons.shape[0] == 4
    if len(segs) <= 1:
        assert cons.shape[1] == 0
        return
    starts, pos = [0], 0
    for prev, cur in zip(segs[:-1], segs[1:]):
        pos += assembly.glue_kernal(cur, prev) if kernal == "glue" else assembly.stick_kernal(cur, prev)
        starts.append(pos)
    length = max(p + len(sg) for p, sg in list(zip(starts, segs))[1:])
    assert cons.shape[1] == length
    cover = np.zeros(length)
    want = np.zeros((4, length))
    for p, sg in zip(starts, segs):
        for j, ch in enumerate(sg):
            if p + j < length:
                cover[p + j] += 1
                want["ACGT".index(ch), p + j] += 1
    assert np.array_equal(cons.sum(axis=0), cover) and np.array_equal(cons, want)
    if kernal == "stick":
        assert ce.index2base(np.argmax(cons, axis=0)) == "".join(segs)
