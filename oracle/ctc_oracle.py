"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- restatement of the two
TensorFlow 1.15 CTC decoder kernels the reference calls
(chiron_eval.py:485-492) and of path_prob (chiron_eval.py:116-136).

The kernels live in tensorflow==1.15.0 (core/kernels/ctc_decoder_ops.cc,
core/util/ctc/ctc_beam_search.h, ctc_beam_entry.h, ctc_loss_util.h), absent
from /root/reference: PARITY UNPINNED against TF itself; validated by
exhaustive path enumeration (brute_force_best) and by the reference's own
pure-Python `mapping()` (easy_assembler.py:26-34) via golden vectors.
"""
import itertools
import math

import numpy as np

NEG_INF = -math.inf


def path_prob(logits):
    """chiron_eval.py:116-136: mean over ALL T frames of (top1 - top2 logit).
    [B,T,K] -> [B,1]."""
    s = np.sort(logits, axis=-1)
    return (s[..., -1] - s[..., -2]).mean(axis=-1, keepdims=True)


def greedy_decode(logits, seq_len, merge_repeated=True):
    """tf.nn.ctc_greedy_decoder (CTCGreedyDecoderOp), batch-major input here.
    Returns (rows: list of int lists, neg_sum_logits [B,1]).
    first-max argmax (Eigen maxCoeff tie rule); blank = K-1."""
    B, T, K = logits.shape
    blank = K - 1
    rows = []
    nsl = np.zeros((B, 1), dtype=logits.dtype)
    for b in range(B):
        prev = -1
        out = []
        acc = logits.dtype.type(0)
        for t in range(int(seq_len[b])):
            k = int(np.argmax(logits[b, t]))
            acc += -logits[b, t, k]
            if k != blank and not (merge_repeated and k == prev):
                out.append(k)
            prev = k
        rows.append(out)
        nsl[b, 0] = acc
    return rows, nsl


def rows_to_sparse(rows, batch):
    """SparseTensor layout produced by both TF decoders: indices [nnz,2]
    (row, position) row-major sorted, values [nnz], dense_shape
    [batch, max_len]."""
    idx, val, mx = [], [], 0
    for b, r in enumerate(rows):
        for j, v in enumerate(r):
            idx.append((b, j))
            val.append(v)
        mx = max(mx, len(r))
    indices = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(val, dtype=np.int64)
    return indices, values, np.asarray([batch, mx], dtype=np.int64)


def _logsumexp2(a, b):
    """ctc_loss_util.h LogSumExp."""
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    if a > b:
        return a + math.log1p(math.exp(b - a))
    return b + math.log1p(math.exp(a - b))


class _Entry(object):
    __slots__ = ("parent", "label", "children", "o_total", "o_blank", "o_label",
                 "n_total", "n_blank", "n_label")

    def __init__(self, parent, label):
        self.parent = parent
        self.label = label
        self.children = None
        self.o_total = self.o_blank = self.o_label = NEG_INF
        self.n_total = self.n_blank = self.n_label = NEG_INF

    def active(self):
        return self.n_total != NEG_INF


def beam_search_decode_row(logits_row, seq_len, beam_width, dtype=np.float64):
    """CTCBeamSearchDecoder<>::Step/TopPaths for one batch row, top_paths=1,
    merge_repeated=False (chiron_eval.py:489-492).  SURVEY appendix A.5.
    Returns (labels, log_prob)."""
    K = logits_row.shape[1]
    blank = K - 1
    root = _Entry(None, -1)
    root.n_total = 0.0
    root.n_blank = 0.0
    root.n_label = NEG_INF
    leaves = [root]
    for t in range(int(seq_len)):
        raw = np.asarray(logits_row[t], dtype=dtype)
        mx = raw.max()
        lse = math.log(float(np.exp(raw - mx).sum()))
        logp = [float(raw[k] - mx) - lse for k in range(K)]
        branches = sorted(leaves, key=lambda e: -e.n_total)
        leaves = []
        for b in branches:
            b.o_total, b.o_blank, b.o_label = b.n_total, b.n_blank, b.n_label
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.o_blank if b.label == b.parent.label else b.parent.o_total
                    b.n_label = _logsumexp2(b.n_label, prev)
                b.n_label += logp[b.label]
            b.n_blank = b.o_total + logp[blank]
            b.n_total = _logsumexp2(b.n_blank, b.n_label)
            leaves.append(b)

        def bottom():
            return min(leaves, key=lambda e: e.n_total)

        def is_candidate(total):
            return total > NEG_INF and (len(leaves) < beam_width or total > bottom().n_total)

        for b in branches:
            if not is_candidate(b.o_total):
                continue
            if b.children is None:
                b.children = [_Entry(b, c) for c in range(K - 1)]
            for c in b.children:
                if c.active():
                    continue
                c.n_blank = NEG_INF
                prev = b.o_blank if c.label == b.label else b.o_total
                c.n_label = logp[c.label] + prev
                c.n_total = c.n_label
                if is_candidate(c.n_total):
                    if len(leaves) == beam_width:
                        bt = bottom()
                        bt.n_total = bt.n_blank = bt.n_label = NEG_INF
                        leaves.remove(bt)
                    leaves.append(c)
                else:
                    c.o_total = c.o_blank = c.o_label = NEG_INF
                    c.n_total = c.n_blank = c.n_label = NEG_INF
    best = max(leaves, key=lambda e: e.n_total)
    labels = []
    e = best
    while e.parent is not None:
        labels.append(e.label)
        e = e.parent
    return labels[::-1], best.n_total


def beam_search_decode(logits, seq_len, beam_width, dtype=np.float64):
    rows, lp = [], []
    for b in range(logits.shape[0]):
        r, p = beam_search_decode_row(logits[b], seq_len[b], beam_width, dtype)
        rows.append(r)
        lp.append(p)
    return rows, np.asarray(lp, dtype=np.float64).reshape(-1, 1)


def collapse(path, blank):
    """CTC many-to-one map B(): merge repeats then drop blanks (same as the
    reference's mapping(), easy_assembler.py:26-34)."""
    out, prev = [], None
    for k in path:
        if k != prev and k != blank:
            out.append(k)
        prev = k
    return tuple(out)


def brute_force_best(logits_row, seq_len):
    """Exhaustive CTC: sum path probabilities per labelling; return the most
    probable labelling and its log-prob.  Exponential -- T <= 7 only."""
    T = int(seq_len)
    K = logits_row.shape[1]
    x = np.asarray(logits_row[:T], dtype=np.float64)
    lp = x - x.max(axis=1, keepdims=True)
    lp = lp - np.log(np.exp(lp).sum(axis=1, keepdims=True))
    table = {}
    for path in itertools.product(range(K), repeat=T):
        s = sum(lp[t, k] for t, k in enumerate(path))
        lab = collapse(path, K - 1)
        table[lab] = np.logaddexp(table.get(lab, -np.inf), s)
    best = max(table, key=lambda k: table[k])
    return list(best), float(table[best]), table
