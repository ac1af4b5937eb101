"""CPU: pin the CTC restatements: greedy vs the reference's own mapping() golden vectors, beam search
vs exhaustive path enumeration (TF's kernel source is unavailable)."""
import numpy as np
import pytest

from oracle import ctc_oracle as co
import regimes


def test_greedy_matches_reference_mapping(golden):
    for case in golden["mapping"]:
        path = case["in"]
        if not path:
            continue
        T = len(path)
        logits = np.full((1, T, 5), -1.0, dtype=np.float32)
        logits[0, np.arange(T), path] = 1.0
        rows, nsl = co.greedy_decode(logits, [T])
        assert rows[0] == case["out"]
        assert nsl[0, 0] == -float(T)


def test_greedy_first_max_tie_rule_and_seq_len():
    lg = np.zeros((2, 4, 5), dtype=np.float32)       # all ties -> class 0 every frame -> merged to one 'A'
    rows, _ = co.greedy_decode(lg, [4, 0])
    assert rows == [[0], []]
    lg[0, 1, 4] = 1.0                                # A - A A  -> blank separates repeats
    rows, _ = co.greedy_decode(lg, [4, 2])
    assert rows[0] == [0, 0] and rows[1] == [0]
    idx, val, shape = co.rows_to_sparse(rows, 2)
    assert idx.tolist() == [[0, 0], [0, 1], [1, 0]] and val.tolist() == [0, 0, 0] and shape.tolist() == [2, 2]


def test_path_prob_all_frames():
    rng = np.random.RandomState(0)
    lg = rng.randn(3, 6, 5).astype(np.float32)
    s = np.sort(lg, axis=-1)
    np.testing.assert_allclose(co.path_prob(lg)[:, 0], (s[..., 4] - s[..., 3]).mean(1), rtol=1e-6)


@pytest.mark.parametrize("seed", range(12))
def test_beam_search_finds_the_exhaustive_optimum(seed):
    rng = np.random.RandomState(seed)
    T = int(rng.randint(1, 7))
    lg = (rng.randn(T, 5) * 2.0).astype(np.float64)
    best, best_lp, table = co.brute_force_best(lg, T)
    labels, lp = co.beam_search_decode_row(lg, T, beam_width=512)   # wide enough to be exact
    assert labels == best
    assert abs(lp - best_lp) < 1e-9
    # every labelling's probability is accounted for: sum over labellings == 1
    assert abs(np.logaddexp.reduce(list(table.values()))) < 1e-9


def test_beam_equals_greedy_on_peaked_posteriors():
    rng = np.random.RandomState(3)
    T = 60
    path = rng.randint(0, 5, T)
    lg = np.full((1, T, 5), -8.0)
    lg[0, np.arange(T), path] = 8.0
    rows, _ = co.greedy_decode(lg, [T])
    b, _ = co.beam_search_decode(lg, [T], 30)
    assert b[0] == rows[0]


def test_beam_width_one_and_seq_len_zero():
    rng = np.random.RandomState(4)
    lg = rng.randn(2, 10, 5)
    rows, lp = co.beam_search_decode(lg, [10, 0], 1)
    assert rows[1] == [] and lp[1, 0] == 0.0
    assert len(rows[0]) <= 10


def test_beam_no_merge_repeated():
    """merge_repeated=False (chiron_eval.py:491): 'A A' separated by blank stays two symbols, and the
    decoder may emit genuine repeats."""
    lg = np.full((1, 3, 5), -9.0)
    lg[0, 0, 0] = lg[0, 1, 4] = lg[0, 2, 0] = 9.0
    rows, _ = co.beam_search_decode(lg, [3], 10)
    assert rows[0] == [0, 0]


@pytest.mark.parametrize("seed", range(6))
def test_c_beam_oracle_matches_python_and_exhaustive(built, seed):
    from oracle import c_oracle
    rng = np.random.RandomState(100 + seed)
    T = int(rng.randint(2, 7))
    lg = (rng.randn(3, T, 5) * 2.0).astype(np.float32)
    sl = np.asarray([T, T - 1, 0])
    rows, lp = c_oracle.beam(lg, sl, 512)
    for b in range(3):
        best, best_lp, _ = co.brute_force_best(lg[b], sl[b])
        assert rows[b] == best
        assert abs(lp[b, 0] - best_lp) < 1e-4
    # narrow beams: identical to the float64 Python restatement (same sequential pruning rules)
    lg = (rng.randn(8, 40, 5) * 2.5).astype(np.float32)
    sl = rng.randint(1, 41, size=8)
    for w in (1, 3, 30):
        r1, l1 = c_oracle.beam(lg, sl, w)
        r2, l2 = co.beam_search_decode(lg, sl, w)
        assert r1 == r2
        np.testing.assert_allclose(l1, l2, rtol=0, atol=1e-4)


def test_decoder_exp_log_pair_accuracy_and_known_bits():
    """The beam-search oracle (and the device decoder, chiron_amd/csrc/ctc_math.h) fix e^x / ln x to a sequence of
    exactly-specified float operations instead of libm.  Here: the pair is accurate to ~1 ulp against float64 (so it is
    a faithful stand-in for the expf / logf / log1pf TF calls), and a handful of outputs are pinned by their bits, so
    any change to the sequence is caught on the CPU before it shows up as a device / oracle mismatch."""
    import struct
    from oracle import c_oracle
    rng = np.random.RandomState(0)
    d = -np.abs(np.concatenate([rng.uniform(0, 86, 4000), rng.uniform(0, 1, 1000), [0.0, 1e-8, 86.0]])).astype(np.float32)
    got = np.array([c_oracle.ctc_exp(v) for v in d], dtype=np.float64)
    want = np.exp(d.astype(np.float64))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - want) / ulp) < 1.5
    assert c_oracle.ctc_exp(-87.0) == 0.0 and c_oracle.ctc_exp(-np.inf) == 0.0 and c_oracle.ctc_exp(0.0) == 1.0
    x = np.concatenate([rng.uniform(1, 2, 3000), rng.uniform(1, 8, 2000), [1.0, 2.0, 5.0]]).astype(np.float32)
    got = np.array([c_oracle.ctc_log(v) for v in x], dtype=np.float64)
    want = np.log(x.astype(np.float64))
    assert np.max(np.abs(got - want)) < 2.5e-7           # ~1 ulp near ln 8, absolute near 1 where ln x -> 0
    assert c_oracle.ctc_log(1.0) == 0.0
    # log-sum-exp: symmetric, -inf is the identity, close to float64
    for a, b in ((-1.0, -2.5), (-30.0, -30.0), (-0.1, -90.0), (-700.0, -701.5)):
        v = c_oracle.ctc_lse(a, b)
        assert v == c_oracle.ctc_lse(b, a)
        assert abs(v - np.logaddexp(a, b)) < 4e-7 * max(1.0, abs(v))
    assert c_oracle.ctc_lse(-np.inf, -3.0) == -3.0 and c_oracle.ctc_lse(-3.0, -np.inf) == -3.0
    bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
    kat = {"exp(-1)": bits(c_oracle.ctc_exp(-1.0)), "exp(-0.3)": bits(c_oracle.ctc_exp(-0.3)),
           "exp(-20.5)": bits(c_oracle.ctc_exp(-20.5)), "log(1.5)": bits(c_oracle.ctc_log(1.5)),
           "log(3.7)": bits(c_oracle.ctc_log(3.7)), "lse(-1,-2.5)": bits(c_oracle.ctc_lse(-1.0, -2.5))}
    assert kat == CTC_MATH_KAT, kat


CTC_MATH_KAT = {"exp(-1)": 1052531378, "exp(-0.3)": 1061004867, "exp(-20.5)": 816566744, "log(1.5)": 1053792543,
                "log(3.7)": 1067939699, "lse(-1,-2.5)": 3209457709}


def test_device_header_restates_the_same_operations(tmp_path):
    """chiron_amd/csrc/ctc_math.h (what beam.hip compiles for the GPU) built for the host with g++ gives, input for
    input, the bits of the oracle's own restatement: the two files describe one function."""
    import ctypes
    import os
    import subprocess
    from oracle import c_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "shim.cpp"
    src.write_text('#include "ctc_math.h"\n'
                   'extern "C" float h_exp(float d) { return ctc_exp_neg(d); }\n'
                   'extern "C" float h_log(float x) { return ctc_log_pos(x); }\n'
                   'extern "C" float h_lse(float a, float b) { return ctc_log_sum_exp(a, b); }\n'
                   'extern "C" void h_lsm(const float* x, float* o) { ctc_log_softmax<5>(x, o); }\n')
    so = tmp_path / "shim.so"
    subprocess.check_call(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-shared", "-fPIC", "-Wno-unknown-pragmas",
                           "-I", os.path.join(root, "chiron_amd", "csrc"), str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    for fn, n in (("h_exp", 1), ("h_log", 1), ("h_lse", 2)):
        getattr(lib, fn).restype = ctypes.c_float
        getattr(lib, fn).argtypes = [ctypes.c_float] * n
    rng = np.random.RandomState(1)
    f32 = lambda v: np.float32(v).view(np.uint32)
    for d in -np.abs(rng.uniform(0, 90, 5000)).astype(np.float32):
        assert f32(lib.h_exp(d)) == f32(c_oracle.ctc_exp(d))
    for x in rng.uniform(1, 8, 5000).astype(np.float32):
        assert f32(lib.h_log(x)) == f32(c_oracle.ctc_log(x))
    for a, b in (-np.abs(rng.randn(5000, 2)) * 40).astype(np.float32):
        assert f32(lib.h_lse(a, b)) == f32(c_oracle.ctc_lse(a, b))
    # the frame log-softmax: header vs oracle through a one-frame beam-1 decode (its log_prob is logp[argmax])
    lib.h_lsm.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    for _ in range(200):
        x = (rng.randn(5) * 3).astype(np.float32)
        o = np.zeros(5, np.float32)
        lib.h_lsm(x.ctypes.data, o.ctypes.data)
        _, lp = c_oracle.beam(x.reshape(1, 1, 5), [1], 1)
        assert f32(lp[0, 0]) == f32(o.max())


def test_beam_scores_against_torch_ctc_forward_algorithm(built):
    """An implementation nobody here wrote: torch.nn.functional.ctc_loss computes -log p(labelling | x) exactly with the CTC
    forward algorithm.  A prefix beam search returns, for its top labelling, the log of the probability mass of that
    labelling's alignments that SURVIVED in the beam -- never more than the exact value, and exactly the exact value while
    no prefix has been pruned (T = 3: 85 prefixes, beam 256).  Checked for the C oracle (float32, the device's bit-exact
    partner) and the float64 Python oracle, on flat and blank-dominated posteriors, ragged lengths, widths 5 .. 256."""
    import torch
    from oracle import c_oracle, ctc_oracle

    def exact_log_prob(lg, sl, rows):
        B, T, K = lg.shape
        lp = torch.log_softmax(torch.from_numpy(lg.astype(np.float64)), dim=2).permute(1, 0, 2)
        tl = torch.tensor([len(r) for r in rows])
        tg = torch.tensor([v for r in rows for v in r], dtype=torch.long)
        loss = torch.nn.functional.ctc_loss(lp, tg, torch.from_numpy(sl.astype(np.int64)), tl, blank=K - 1, reduction="none")
        return -loss.numpy()

    rng = np.random.RandomState(3)
    for scale, bias in ((0.7, 0.0), (4.0, 3.0)):
        for T, W in ((3, 256), (10, 256), (60, 30), (400, 30), (400, 5)):
            B = 12
            lg = (rng.randn(B, T, 5) * scale).astype(np.float32)
            lg[..., 4] += bias
            sl = rng.randint(1, T + 1, size=B).astype(np.int32)
            sl[0] = T
            rows, lp = c_oracle.beam(lg, sl, W)
            ex = exact_log_prob(lg, sl, rows)
            assert (lp[:, 0] <= ex + 2e-5 * np.maximum(1.0, np.abs(ex))).all(), (T, W, scale)
            if T == 3:
                assert np.abs(lp[:, 0] - ex).max() < 1e-5
            if T <= 60:   # the float64 Python restatement (slow): the same bound, tighter tolerance
                prow, plp = ctc_oracle.beam_search_decode(lg, sl, W)
                pex = exact_log_prob(lg, sl, prow)
                assert (plp[:, 0] <= pex + 1e-9).all()
                if T == 3:
                    assert np.abs(plp[:, 0] - pex).max() < 1e-9


def test_greedy_base_count_equals_the_oracle_decode_lengths():
    """regimes.greedy_base_count (used by fit_emitting_head to bisect a head's blank bias, and by bench.py's realistic-density leg) counts
    what tf.nn.ctc_greedy_decoder emits: the oracle's greedy decode, row by row, ragged lengths included."""
    import chiron_amd as ca
    rng = np.random.RandomState(3)
    for _ in range(20):
        B, T = int(rng.randint(1, 9)), int(rng.randint(1, 40))
        lg = rng.normal(0, 2, (B, T, 5)).astype(np.float32)
        lg[..., 4] += rng.choice([-2.0, 0.0, 3.0])
        sl = rng.randint(0, T + 1, size=B)
        rows, _ = co.greedy_decode(lg, sl)
        assert regimes.greedy_base_count(lg, sl).tolist() == [len(r) for r in rows]


def test_fit_emitting_head_reaches_the_requested_density():
    """regimes.fit_emitting_head: a recurrent output that encodes the squiggle's level changes (here: planted, plus noise) under the fitted
    head decodes the requested number of bases per window, on the calibration windows and on held-out ones; only the four head tensors
    of the weight dict change."""
    import chiron_amd as ca
    rng = np.random.RandomState(11)
    B, T, H = 24, 200, 100
    x = np.zeros((B, T), dtype=np.float32)
    for b in range(B):                                  # a squiggle: a new level every ~9 samples
        t = 0
        while t < T:
            n = int(rng.geometric(1.0 / 9.0))
            x[b, t:t + n] = rng.normal(500, 80)
            t += n
    x += rng.normal(0, 4, x.shape).astype(np.float32)
    jump = np.abs(np.diff(x, axis=1, prepend=x[:, :1])) > 32
    q = np.searchsorted(np.quantile(x, [0.25, 0.5, 0.75]), x)
    proj = rng.normal(0, 1, (6, 2 * H))
    feat = np.stack([jump * (q == k) for k in range(4)] + [~jump, np.ones_like(jump)], axis=-1).astype(np.float64)
    h = np.tanh(feat @ proj + rng.normal(0, 0.05, (B, T, 2 * H)))
    w = {"rnn_fnn_layer/weights": np.zeros((2, H), np.float32), "rnn_fnn_layer/bias": np.zeros(H, np.float32),
         "rnn_fnn_layer/weights_class": np.zeros((H, 5), np.float32), "rnn_fnn_layer/bias_class": np.zeros(5, np.float32), "other": np.ones(3)}
    sl = np.full(B, T)
    out = regimes.fit_emitting_head(w, h[:16], x[:16], sl[:16], 12.0, hidden=H)
    assert out["other"] is w["other"] and set(out) == set(w)

    def density(rows):
        pre = (h[rows, :, :H] * out["rnn_fnn_layer/weights"][0] + h[rows, :, H:] * out["rnn_fnn_layer/weights"][1]) + out["rnn_fnn_layer/bias"]
        lg = pre @ out["rnn_fnn_layer/weights_class"] + out["rnn_fnn_layer/bias_class"]
        return regimes.greedy_base_count(lg, sl[rows]).mean()

    assert abs(density(slice(0, 16)) - 12.0) < 0.5 and abs(density(slice(16, 24)) - 12.0) < 3.0
