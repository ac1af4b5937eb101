"""CPU: pin the numpy / C restatements of the network against independent implementations
(torch CPU conv1d / LSTMCell), hand-checked TF padding arithmetic and each other.
The real TF 1.15 path cannot run here: NN parity vs TF is UNPINNED."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import chiron_amd as ca
from oracle import nn_oracle as no


def test_same_padding_table():
    # k=3,s=1 -> 1/1; k=13,s=5,W=500 -> 4/4; k=14,s=7,W=500 -> 5/6; 1x1 s=5 -> none
    assert no.same_padding(400, 3, 1) == (400, 1, 1)
    assert no.same_padding(500, 13, 5) == (100, 4, 4)
    assert no.same_padding(500, 14, 7) == (72, 5, 6)
    assert no.same_padding(500, 1, 5) == (100, 0, 0)
    assert no.same_padding(2000, 13, 5) == (400, 4, 4)
    assert no.same_padding(7, 3, 2) == (4, 1, 1)
    assert no.same_padding(8, 3, 2) == (4, 0, 1)


@pytest.mark.parametrize("k,stride,W", [(1, 1, 17), (3, 1, 40), (13, 5, 103), (13, 5, 500), (14, 7, 100), (1, 5, 23), (3, 2, 8)])
def test_conv_matches_torch(k, stride, W):
    rng = np.random.RandomState(k * 100 + stride)
    B, ci, co = 3, 5, 7
    x = rng.randn(B, W, ci)
    w = rng.randn(k, ci, co)
    y = no.conv1d_same(x, w, stride)
    out, left, right = no.same_padding(W, k, stride)
    xt = F.pad(torch.from_numpy(x).permute(0, 2, 1), (left, right))
    yt = F.conv1d(xt, torch.from_numpy(w).permute(2, 1, 0).contiguous(), stride=stride)
    assert yt.shape[-1] == out
    np.testing.assert_allclose(y, yt.permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)


def _torch_lstm_dir(x, seq_len, kernel, bias, reverse):
    """Independent second opinion with torch.nn.LSTMCell (gate order i,f,g,o) and explicit masking."""
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    cell = torch.nn.LSTMCell(D, H).double()
    i, j, f, o = [kernel[:, q * H:(q + 1) * H] for q in range(4)]
    bi, bj, bf, bo = [bias[q * H:(q + 1) * H] for q in range(4)]
    Wt = np.concatenate([i, f, j, o], axis=1)            # torch order: i, f, g(=j), o
    bt = np.concatenate([bi, bf + 1.0, bj, bo])          # forget_bias 1.0
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(Wt[:D].T))
        cell.weight_hh.copy_(torch.from_numpy(Wt[D:].T))
        cell.bias_ih.copy_(torch.from_numpy(bt))
        cell.bias_hh.zero_()
    out = np.zeros((B, T, H))
    for b in range(B):
        n = int(seq_len[b])
        xs = torch.from_numpy(x[b, :n])
        if reverse:
            xs = torch.flip(xs, [0])
        h = torch.zeros(1, H, dtype=torch.float64)
        c = torch.zeros(1, H, dtype=torch.float64)
        ys = []
        with torch.no_grad():
            for t in range(n):
                h, c = cell(xs[t:t + 1], (h, c))
                ys.append(h[0].numpy().copy())
        ys = np.asarray(ys).reshape(n, H)
        if reverse:
            ys = ys[::-1]
        out[b, :n] = ys
    return out


@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_direction_matches_torch(reverse):
    rng = np.random.RandomState(5)
    B, T, D, H = 4, 9, 6, 5
    x = rng.randn(B, T, D)
    kernel = rng.randn(D + H, 4 * H) * 0.5
    bias = rng.randn(4 * H) * 0.3
    seq = np.asarray([9, 4, 1, 0])
    got = no.lstm_direction(x, seq, kernel, bias, reverse)
    want = _torch_lstm_dir(x, seq, kernel, bias, reverse)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    # masked frames read back as exact zeros
    assert np.all(got[1, 4:] == 0) and np.all(got[3] == 0)


def test_reverse_sequence():
    x = np.arange(2 * 5 * 1, dtype=np.float64).reshape(2, 5, 1)
    y = no.reverse_sequence(x, [3, 5])
    assert y[0, :, 0].tolist() == [2, 1, 0, 3, 4]
    assert y[1, :, 0].tolist() == [9, 8, 7, 6, 5]


def test_multi_equals_manual_stack():
    """rnn.py:99-174 MultiRNNCell form: a direction's layer l+1 sees only that direction's layer l."""
    spec = ca.rna_default_spec()
    w = ca.synthetic_weights(spec, seed=3)
    rng = np.random.RandomState(0)
    fea = rng.randn(2, 7, 256)
    seq = [7, 3]
    out = no.rnn_forward(fea, seq, spec.to_dict(), {k: np.asarray(v, np.float64) for k, v in w.items()})
    x = fea
    for l in range(3):
        p = "BDGRU_rnn/fw/multi_rnn_cell/cell_%d/lstm_cell/" % l
        x = no.lstm_direction(x, seq, w[p + "kernel"].astype(np.float64), w[p + "bias"].astype(np.float64), False)
    np.testing.assert_allclose(out[:, :, :100], x, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("kind", ["stack", "multi"])
def test_layer_stages_compose_to_the_whole_stack(kind):
    """nn_oracle.rnn_layer_forward (one recurrent layer as a stage: the per-stage error budget of tools/parity_budget.py applies
    it to an implementation's own previous-stage output) composed over the layers == rnn_forward, bit for bit, float64 and
    float32, ragged lengths including 0 and 1 -- for the MultiRNN graph too, where the backward stack is written as
    reverse / 3 layers / reverse in rnn_forward and as a reversal around every layer here."""
    import chiron_amd as ca
    spec = ca.dna_default_spec() if kind == "stack" else ca.rna_default_spec()
    sd = spec.to_dict()
    w = ca.synthetic_weights(spec, seed=3, lstm_gain=2.0)
    rng = np.random.RandomState(1)
    fea = rng.randn(6, 23, 256)
    sl = np.array([23, 11, 0, 1, 20, 23])
    for dt in (np.float64, np.float32):
        ww = {k: np.asarray(v, dtype=dt) for k, v in w.items()}
        x = fea.astype(dt)
        ref = no.rnn_forward(x, sl, sd, ww)
        for l in range(3):
            x = no.rnn_layer_forward(x, sl, sd, ww, l)
        assert x.dtype == dt and np.array_equal(x, ref)


def test_fc_head_formula():
    rng = np.random.RandomState(1)
    H, K = 100, 5
    lasth = rng.randn(2, 3, 2 * H)
    w = {"rnn_fnn_layer/weights": rng.randn(2, H), "rnn_fnn_layer/bias": rng.randn(H),
         "rnn_fnn_layer/weights_class": rng.randn(H, K), "rnn_fnn_layer/bias_class": rng.randn(K)}
    got = no.fc_head(lasth, w)
    v = lasth[:, :, :H] * w["rnn_fnn_layer/weights"][0] + lasth[:, :, H:] * w["rnn_fnn_layer/weights"][1] + w["rnn_fnn_layer/bias"]
    np.testing.assert_allclose(got, v @ w["rnn_fnn_layer/weights_class"] + w["rnn_fnn_layer/bias_class"], rtol=1e-12)


def test_bn_modes():
    rng = np.random.RandomState(2)
    x = rng.randn(3, 11, 4) * 3 + 1
    w = {"s_bn/scale": rng.rand(4) + 0.5, "s_bn/offset": rng.randn(4), "s_bn/pop_mean": rng.randn(4), "s_bn/pop_var": rng.rand(4) + 0.1}
    pop = no.bn_site(x, w, "s", "population")
    np.testing.assert_allclose(pop, (x - w["s_bn/pop_mean"]) / np.sqrt(w["s_bn/pop_var"] + no.BN_EPS) * w["s_bn/scale"] + w["s_bn/offset"], rtol=1e-12)
    bat = no.bn_site(x, w, "s", "batch")     # HEAD simple_global_bn: moments over [0,1,2], biased var
    m, v = x.reshape(-1, 4).mean(0), x.reshape(-1, 4).var(0)
    np.testing.assert_allclose(bat, (x - m) / np.sqrt(v + no.BN_EPS) * w["s_bn/scale"] + w["s_bn/offset"], rtol=1e-12)


@pytest.mark.parametrize("kind", ["dna", "rna"])
def test_c_oracle_matches_numpy_oracle(built, kind):
    from oracle import c_oracle, ctc_oracle
    spec, L = (ca.dna_default_spec(), 400) if kind == "dna" else (ca.rna_default_spec(), 500)
    w = ca.synthetic_weights(spec, seed=11)
    B = 5
    x = ca.synthetic_signal(1, B * L, seed=2)[0].reshape(B, L)
    T = spec.output_len(L)
    seq = np.asarray([T, T, T // 2, 1, 0])
    ref, ratio = no.inference(x, seq, spec.to_dict(), w, dtype=np.float64)
    assert ratio == L / T
    got = c_oracle.forward(x, seq, spec.to_dict(), spec.pack(w), T)
    assert np.abs(got - ref).max() < 5e-5
    r1, n1 = ctc_oracle.greedy_decode(got, seq)
    r2, n2, pp = c_oracle.greedy(got, seq)
    assert r1 == r2
    np.testing.assert_allclose(n1, n2, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pp, ctc_oracle.path_prob(got), rtol=1e-5, atol=1e-5)


def test_weight_manifest_matches_checkpoint_index_shapes():
    """68 inference tensors, ~1.83 M parameters for DNA_default."""
    spec = ca.dna_default_spec()
    v = spec.variables()
    assert len(v) == 68
    assert v["BDLSTM_rnn/cell_0/bidirectional_rnn/fw/lstm_cell/kernel"] == (356, 400)
    assert v["BDLSTM_rnn/cell_2/bidirectional_rnn/bw/lstm_cell/kernel"] == (300, 400)
    assert v["res_layer2/branch2/conv2b/weights"] == (1, 3, 256, 256)
    assert "res_layer2/branch1/conv1_bn/scale" not in v and "res_layer1/branch1/conv1_bn/pop_var" in v
    n = sum(int(np.prod(s)) for s in v.values())
    assert n == 1827333
    r = ca.rna_default_spec().variables()
    assert r["res_layer1/branch2/conv2b/weights"] == (1, 13, 256, 256)
    assert r["BDGRU_rnn/fw/multi_rnn_cell/cell_1/lstm_cell/kernel"] == (200, 400)
    assert r["BDGRU_rnn/bw/multi_rnn_cell/cell_0/lstm_cell/kernel"] == (356, 400)
