"""CPU: pin the CTC restatements: greedy vs the reference's own mapping() golden vectors, beam search
vs exhaustive path enumeration (TF's kernel source is unavailable)."""
import numpy as np
import pytest

from oracle import ctc_oracle as co


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
