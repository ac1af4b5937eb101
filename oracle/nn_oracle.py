"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of
the reference network: chiron/cnn.py residual stack, chiron/rnn.py BiLSTM and
FC head, exactly as composed in the shipped graphs
(chiron/model/*/final.ckpt-*.meta) and TF 1.15 op semantics.

PINNING: tensorflow==1.15.0 and the trained weights are not available here
(SURVEY.md section 0, facts 1-3), so no TF-produced logits exist to compare
with.  What IS reference-held is the trained graph itself: the MetaGraphDefs
under chiron/model/*/ record every op of the network and how they are wired.
tests/golden/make_meta_golden.py executes those node lists (tf.cond as
Switch/Merge, dynamic_rnn as while-loop frames with TensorArrays) in float64 on
seeded inputs at the graphs' own shapes, and tests/test_meta_golden.py holds this
file to those activations at 1e-12 (population BN for DNA and RNA, the
batch-statistics branch for DNA) and to the structural digest
tests/golden/meta_graph.json.  The composition is therefore pinned to the
reference; the arithmetic inside each primitive op (SAME-padded Conv2D, Sigmoid,
Tanh, MatMul accumulation order) remains TF-kernel behaviour that can only be
restated -- cross-checked against torch CPU in tests/test_oracle_nn.py.

Inputs are plain dicts so that this file depends on nothing in the product:

spec = {"cnn": [{"name": "res_layer1", "in": 1, "out": 256, "k": 3,
                 "stride": 1, "i_bn": True}, ...],
        "rnn": {"kind": "stack" | "multi", "layers": 3, "hidden": 100},
        "bn_mode": "population" | "batch", "classes": 5}
weights = {tf_variable_name: ndarray} with the names/shapes of the shipped
checkpoint index.
"""
import math

import numpy as np

# cnn.py:125 (epsilon=1e-5), cnn.py:188.  The graphs are float32, so the constant the reference's BN adds is
# float32(1e-5) = 9.999999747378752e-06 (the value of .../batchnorm_1/add/y in the shipped .meta, tests/golden/meta_graph.json);
# the float64 oracle uses that same number, not the double 1e-5.
BN_EPS = float(np.float32(1e-5))
FORGET_BIAS = 1.0      # tf LSTMCell default forget_bias, const 1.0 in .meta


def same_padding(width, k, stride):
    """TF 'SAME' padding along W (cnn.py:60-64 -> tf.nn.conv2d padding=SAME).

    out = ceil(W/s); pad_total = max((out-1)*s + k - W, 0); left = total//2.
    """
    out = -(-width // stride)
    pad_total = max((out - 1) * stride + k - width, 0)
    left = pad_total // 2
    return out, left, pad_total - left


def conv1d_same(x, w, stride=1):
    """x [B,W,Cin], w [k,Cin,Cout] (TF HWIO with H=1 squeezed), cross-
    correlation (no flip), SAME padding.  cnn.py:60-64."""
    B, W, Cin = x.shape
    k, Cin2, Cout = w.shape
    assert Cin == Cin2
    out, left, right = same_padding(W, k, stride)
    xp = np.zeros((B, W + left + right, Cin), dtype=x.dtype)
    xp[:, left:left + W] = x
    y = np.zeros((B, out, Cout), dtype=x.dtype)
    for tap in range(k):
        xs = xp[:, tap:tap + (out - 1) * stride + 1:stride]      # [B,out,Cin]
        y += xs @ w[tap]
    return y


def bn_apply(x, scale, offset, mean, var):
    """tf.nn.batch_normalization association order (.meta batchnorm_1/*):
    inv = rsqrt(var+eps)*scale; y = x*inv + (offset - mean*inv)."""
    inv = (1.0 / np.sqrt(var + x.dtype.type(BN_EPS))) * scale
    return x * inv + (offset - mean * inv)


def bn_site(x, weights, site, mode):
    """population: cnn.py:125-163 batchnorm() inference branch (shipped
    checkpoints).  batch: cnn.py:166-188 simple_global_bn (HEAD) -- moments
    over axes [0,1,2] of THIS batch, biased variance."""
    if mode == "population":
        return bn_apply(x, weights[site + "_bn/scale"], weights[site + "_bn/offset"],
                        weights[site + "_bn/pop_mean"], weights[site + "_bn/pop_var"])
    mean = x.mean(axis=(0, 1))
    var = ((x - mean) ** 2).mean(axis=(0, 1))
    return bn_apply(x, weights[site + "_bn/scale"], weights[site + "_bn/offset"], mean, var)


def conv_layer(x, weights, site, stride, bn, relu, bn_mode):
    """cnn.py:15-83 restricted to what the shipped graphs use (no bias,
    no dilation, relu only)."""
    w = weights[site + "/weights"]
    w = w.reshape(w.shape[-3], w.shape[-2], w.shape[-1])        # drop H=1
    y = conv1d_same(x, w.astype(x.dtype), stride)
    if bn:
        y = bn_site(y, weights, site, bn_mode)
    if relu:
        y = np.maximum(y, 0)
    return y


def residual_layer(x, weights, blk, bn_mode):
    """cnn.py:234-262.  branch1: 1x1 conv (stride) + BN iff i_bn, no act.
    branch2: 1x1+BN+ReLU -> 1xk(stride)+BN+ReLU -> 1x1+BN.  relu(b1+b2)."""
    n = blk["name"]
    s = blk.get("stride", 1)
    b1 = conv_layer(x, weights, n + "/branch1/conv1", s, blk["i_bn"], False, bn_mode)
    a = conv_layer(x, weights, n + "/branch2/conv2a", 1, True, True, bn_mode)
    b = conv_layer(a, weights, n + "/branch2/conv2b", s, True, True, bn_mode)
    c = conv_layer(b, weights, n + "/branch2/conv2c", 1, True, False, bn_mode)
    return np.maximum(b1 + c, 0)


def cnn_forward(signal, spec, weights):
    """cnn.py:334-371 getcnnfeature: [B,L] -> [B,T,C]."""
    x = signal[:, :, None]
    if spec.get("stem"):
        # HEAD RNA_model2 / RNA_model3 (cnn.py:454-476): conv_layer(net, [1,k,1,C], SAME, strides=s) + BN + ReLU
        x = conv_layer(x, weights, "conv_layer/conv1", spec["stem"]["stride"], True, True, spec["bn_mode"])
    for blk in spec["cnn"]:
        x = residual_layer(x, weights, blk, spec["bn_mode"])
    return x


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def lstm_direction(x, seq_len, kernel, bias, reverse):
    """TF LSTMCell inside dynamic_rnn with sequence_length (SURVEY appendix
    A.2): z=[x_t,h]@kernel+bias; i,j,f,o=split(z,4);
    c=sigmoid(f+1)*c+sigmoid(i)*tanh(j); h=sigmoid(o)*tanh(c); frames
    t>=seq_len[b] emit 0 and carry state.  reverse: ReverseSequence on the
    first seq_len[b] frames before and after (rnn.py:64 via
    stack_bidirectional_dynamic_rnn / bidirectional_dynamic_rnn)."""
    B, T, _ = x.shape
    H = kernel.shape[1] // 4
    dt = x.dtype
    kernel = kernel.astype(dt)
    bias = bias.astype(dt)
    out = np.zeros((B, T, H), dtype=dt)
    h = np.zeros((B, H), dtype=dt)
    c = np.zeros((B, H), dtype=dt)
    seq_len = np.asarray(seq_len).astype(np.int64)
    rows = np.arange(B)
    for step in range(T):
        active = step < seq_len
        if not active.any():
            break
        if reverse:
            t_idx = np.where(active, seq_len - 1 - step, 0)
        else:
            t_idx = np.full(B, step)
        xt = x[rows, t_idx]
        z = np.concatenate([xt, h], axis=1) @ kernel + bias
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c_new = _sigmoid(f + dt.type(FORGET_BIAS)) * c + _sigmoid(i) * np.tanh(j)
        h_new = _sigmoid(o) * np.tanh(c_new)
        m = active[:, None]
        c = np.where(m, c_new, c)
        h = np.where(m, h_new, h)
        out[rows[active], t_idx[active]] = h_new[active]
    return out


def rnn_forward(fea, seq_len, spec, weights):
    """rnn.py:20-97 ('stack': DNA, concat fw/bw after every layer) and
    rnn.py:99-174 ('multi': RNA, 3-deep fw stack and bw stack, one concat)."""
    r = spec["rnn"]
    L = r["layers"]
    if r["kind"] == "stack":
        x = fea
        for l in range(L):
            outs = []
            for d, rev in (("fw", False), ("bw", True)):
                p = "BDLSTM_rnn/cell_%d/bidirectional_rnn/%s/lstm_cell/" % (l, d)
                outs.append(lstm_direction(x, seq_len, weights[p + "kernel"], weights[p + "bias"], rev))
            x = np.concatenate(outs, axis=2)
        return x
    outs = []
    for d, rev in (("fw", False), ("bw", True)):
        x = fea
        if rev:
            x = reverse_sequence(x, seq_len)
        for l in range(L):
            p = "BDGRU_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell/" % (d, l)
            x = lstm_direction(x, seq_len, weights[p + "kernel"], weights[p + "bias"], False)
        if rev:
            x = reverse_sequence(x, seq_len)
        outs.append(x)
    return np.concatenate(outs, axis=2)


def rnn_layer_forward(x, seq_len, spec, weights, layer):
    """ONE layer of the recurrent stack as a stage of its own: x = the previous stage's output ([B,T,C] features for layer 0,
    [B,T,2H] after that) -> [B,T,2H].  'stack' (rnn.py:63-65): both directions read the whole x.  'multi' (rnn.py:140-145):
    above layer 0 the forward cell reads x[..., :H] and the backward cell x[..., H:]; the backward stack lives in the
    reversed domain, and reversing a layer's output and the next layer's input again is the identity on the first seq_len
    frames (the rest is zero either way), so composing these stages equals rnn_forward bit for bit (tests/test_oracle_nn.py).
    The per-stage error budget (tools/parity_budget.py) applies this float64 stage to an implementation's own previous-stage
    output: what remains is the error BORN in the stage."""
    r = spec["rnn"]
    H = r["hidden"]
    outs = []
    for di, (d, rev) in enumerate((("fw", False), ("bw", True))):
        if r["kind"] == "stack":
            p = "BDLSTM_rnn/cell_%d/bidirectional_rnn/%s/lstm_cell/" % (layer, d)
            xin = x
        else:
            p = "BDGRU_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell/" % (d, layer)
            xin = x if layer == 0 else x[:, :, di * H:(di + 1) * H]
        outs.append(lstm_direction(np.ascontiguousarray(xin), seq_len, weights[p + "kernel"], weights[p + "bias"], rev))
    return np.concatenate(outs, axis=2)


def reverse_sequence(x, seq_len):
    """tf.reverse_sequence(seq_dim=1, batch_dim=0): only the first seq_len[b]
    frames of each row are reversed, the tail stays in place."""
    y = x.copy()
    for b, n in enumerate(np.asarray(seq_len).astype(np.int64)):
        y[b, :n] = x[b, :n][::-1]
    return y


def fc_head(lasth, weights):
    """rnn.py:72-96: [B,T,2,H]*weights[2,H] summed over
    the 2, +bias[H], @weights_class[H,K] + bias_class[K]."""
    B, T, H2 = lasth.shape
    H = H2 // 2
    dt = lasth.dtype
    w = weights["rnn_fnn_layer/weights"].astype(dt)
    v = lasth.reshape(B, T, 2, H) * w
    v = v.sum(axis=2) + weights["rnn_fnn_layer/bias"].astype(dt)
    return v @ weights["rnn_fnn_layer/weights_class"].astype(dt) + weights["rnn_fnn_layer/bias_class"].astype(dt)


def inference(signal, seq_len, spec, weights, dtype=np.float64, return_all=False):
    """chiron_model.py:134-172 inference(): -> (logits [B,T,K], ratio).
    seq_len is already divided by ratio by the caller (chiron_eval.py:337)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    x = np.asarray(signal, dtype=dtype)
    fea = cnn_forward(x, spec, w)
    ratio = signal.shape[1] / fea.shape[1]                      # chiron_model.py:151-152
    lasth = rnn_forward(fea, seq_len, spec, w)
    logits = fc_head(lasth, w)
    if return_all:
        return logits, ratio, fea, lasth
    return logits, ratio


def output_len(segment_len, spec):
    t = segment_len
    for blk in spec["cnn"]:
        t = math.ceil(t / blk.get("stride", 1))
    return t
