"""Weight regimes for the parity tests (test infrastructure; uses the float64 oracle to calibrate BN statistics).

ca.synthetic_weights keeps every BN site near N(0, 1) and the LSTM gates in their linear region -- convenient, and not what
a trained checkpoint looks like.  Two families are built here on top of it:

* saturated_gate_cases: gate biases of +-50 .. +-120 (cell written through / closed / integrating without output / held and
  output) and a 1e-3 kernel gain (every gate at its midpoint): rnn.py:45-65 LSTMCell, SURVEY A.2.
* trained_like_weights: convolution filters with heterogeneous input-channel scales (log-normal, two decades) and a non-zero
  mean per output channel, BN scale in +-[0.3, 3] and offsets of order 1 (post-BN activations reach +-10 and beyond), population
  statistics CALIBRATED on data the way training's moving averages are (the float64 oracle walks the network and records every
  site's moments -- so conv outputs have large means that the folded BN shift cancels, the case where a re-associated
  convolution such as Winograd F(4,3) with its x8 and /24 constants loses most), LSTM gate biases spread over several units
  so that a good share of the gates saturates per step while the dynamics stay contracting.
"""
import numpy as np

import chiron_amd as ca

H = 100

SATURATED = {"write-through": (50.0, -50.0, 50.0), "closed": (-50.0, -50.0, -50.0), "integrate-no-output": (120.0, 80.0, -90.0),
             "hold-and-output": (-50.0, 50.0, 50.0), "midpoint": "tiny"}


def saturated_gate_weights(spec, name, seed=9):
    """(i, f, o) biases.  All three open at once is left out on purpose: the cell then integrates hundreds of steps of +-1 through
    a recurrent loop of gain ~1 and fp32 / fp64 restatements of the SAME formulas drift apart by 0.5 (numpy oracle at both
    precisions): sensitivity of the network, not of an implementation."""
    gates = SATURATED[name]
    if gates == "tiny":
        return ca.synthetic_weights(spec, seed=seed, lstm_gain=1e-3)
    w = ca.synthetic_weights(spec, seed=seed)
    h = spec.hidden
    for k in w:
        if k.endswith("lstm_cell/bias"):
            b = w[k]                     # columns i | j | f | o
            b[0:h], b[2 * h:3 * h], b[3 * h:4 * h] = gates
    return w


def trained_like_weights(spec, x_calib, seed=5, lstm_gain=3.0, gate_spread=3.0, chan_sigma=1.0, filter_mean=0.6, forget_mean=1.0):
    """forget_mean: centre of the forget-gate biases (TF adds its forget_bias 1.0 on top).  1.0 = long memories (outputs change every
    ~15 frames); -2.0 = cells that follow their input, as a basecaller's must to emit a base every 9 samples (dense_head)."""
    from oracle import nn_oracle
    w = ca.synthetic_weights(spec, seed=seed, lstm_gain=lstm_gain)
    rng = np.random.RandomState(seed + 1000)
    for k in list(w):
        a = w[k]
        if k.endswith("/weights") and a.ndim == 4:
            _, kk, cin, cout = a.shape
            if cin > 1:
                a = a * np.exp(rng.normal(0.0, chan_sigma, (1, 1, cin, 1)))
                if "/branch1/" not in k:           # the shortcut of blocks 2, 3 has no BN behind it: leave its mean alone
                    a = a + rng.normal(0.0, filter_mean / np.sqrt(kk * cin), (1, 1, 1, cout))
                else:
                    a = a * np.exp(-0.5 * chan_sigma * chan_sigma) * 0.5
            w[k] = a.astype(np.float32)
        elif k.endswith("_bn/scale"):
            w[k] = (rng.uniform(0.3, 3.0, a.shape) * rng.choice([-1.0, 1.0], a.shape, p=[0.15, 0.85])).astype(np.float32)
        elif k.endswith("_bn/offset"):
            w[k] = rng.normal(0.0, 1.0, a.shape).astype(np.float32)
        elif k.endswith("lstm_cell/bias"):
            h = spec.hidden
            b = np.empty(4 * h)
            b[0:h] = rng.normal(0.0, gate_spread, h)           # i
            b[h:2 * h] = rng.normal(0.0, 0.5, h)               # j
            b[2 * h:3 * h] = rng.normal(forget_mean, gate_spread, h)   # f (+1.0 forget bias on top)
            b[3 * h:4 * h] = rng.normal(0.0, gate_spread, h)   # o
            w[k] = b.astype(np.float32)
    # population statistics = the moments the data really has at each site (float64 walk, upstream sites already calibrated)
    orig = nn_oracle.bn_site

    def calibrating(xx, weights, site, mode):
        weights[site + "_bn/pop_mean"] = xx.mean(axis=(0, 1)).astype(np.float32)
        weights[site + "_bn/pop_var"] = xx.var(axis=(0, 1)).astype(np.float32)
        return orig(xx, weights, site, "population")

    nn_oracle.bn_site = calibrating
    try:
        fea = nn_oracle.cnn_forward(np.asarray(x_calib, dtype=np.float64), spec.to_dict(), w)
    finally:
        nn_oracle.bn_site = orig
    # the LSTM sees features of order 1 in synthetic_weights; keep its input projection in the same range
    s = float(np.sqrt((fea ** 2).mean()))
    for k in w:
        if k.endswith("lstm_cell/kernel") and ("cell_0" in k):
            nin = w[k].shape[0] - spec.hidden
            w[k][:nin] = (w[k][:nin] / max(s, 1e-6) * 0.75).astype(np.float32)
    return w, fea


def peaked_head(w, gain=4.0, blank_bias=2.0):
    """Trained-CTC-like posteriors on top of any weight set: class layer x gain and the blank ahead by default, so that most
    frames are decided by a wide margin (blank, or one base on evidence) -- as a trained model's are."""
    w = dict(w)
    w["rnn_fnn_layer/weights_class"] = (w["rnn_fnn_layer/weights_class"] * gain).astype(np.float32)
    bc = w["rnn_fnn_layer/bias_class"].copy()
    bc[4] += blank_bias
    w["rnn_fnn_layer/bias_class"] = bc
    return w


def greedy_base_count(logits, seq_len):
    """bases per window of tf.nn.ctc_greedy_decoder (merge_repeated=True, blank = last class; chiron_eval.py:485-487): a frame emits
    its argmax iff that is not the blank and not the previous frame's argmax.  Counting only -- the decode itself is the engine's."""
    am = np.argmax(logits, axis=-1)
    T = am.shape[1]
    valid = np.arange(T)[None, :] < np.asarray(seq_len)[:, None]
    prev = np.concatenate([np.full((am.shape[0], 1), -1, dtype=am.dtype), am[:, :-1]], axis=1)
    return ((am != logits.shape[-1] - 1) & (am != prev) & valid).sum(axis=1)


def fit_emitting_head(weights, lasth, x, seq_len, bases_per_window, hidden=100, ridge=1e-3, logit_scale=8.0):
    """Synthetic weights decode a handful of bases per window; a trained Chiron model emits 16 .. 45 per 400 samples
    (chiron/example_data/DNA/output/segments) -- 43.9 at 450 bases/s and 4 kHz.  This FITS the FC head
    (rnn.py:72-96: rnn_fnn_layer/*) of a weight set to emit like one: per-frame targets from the signal (a base where the squiggle
    jumps by more than four noise sigmas, the base = the quartile of the new level, blank elsewhere; with a strided CNN at most
    every second frame), a class-balanced ridge regression from `lasth` [B, T, 2H] (the recurrent stack's output on the windows x:
    chiron_engine_rnn_output, or an oracle's) through the head's own parameterisation with rnn_fnn_layer/weights = 1, logits scaled
    to `logit_scale` (decided frames are decided by a wide margin, as a trained CTC head's), and the blank bias bisected until the
    greedy decode of these windows holds `bases_per_window`.  Returns a new weight dict; everything but the four head tensors is
    shared.  How fast a head CAN switch is the recurrent stack's business: cells that follow their input (forget-gate biases
    around -2) reach 45 bases per 400 frames, long memories about 25."""
    h = np.asarray(lasth, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    B, T, _ = h.shape
    stride = x.shape[1] // T
    jump = np.abs(np.diff(x, axis=1, prepend=x[:, :1])) > 32.0
    jump = jump[:, :T * stride].reshape(B, T, stride).any(axis=2)
    level = x[:, :T * stride].reshape(B, T, stride)[:, :, -1]
    base = np.searchsorted(np.quantile(level, [0.25, 0.5, 0.75]), level)
    lab = np.where(jump, base, 4)
    if stride > 1:
        lab[:, 1::2] = 4          # a blank between two bases (what CTC needs for a repeat)
    valid = np.arange(T)[None, :] < np.asarray(seq_len)[:, None]
    H = hidden
    pre_all = h[..., :H] + h[..., H:]
    pre = pre_all[valid]
    Y = np.eye(5)[lab[valid]]
    sw = (Y / np.maximum(Y.mean(0), 1e-6)).sum(1)
    mu = pre.mean(0)
    pc = pre - mu
    A = (pc * sw[:, None]).T @ pc
    A += ridge * np.trace(A) / H * np.eye(H)
    Wc = np.linalg.solve(A, (pc * sw[:, None]).T @ (Y - Y.mean(0)))
    gain = logit_scale / max(float(np.abs(pc @ Wc).std()), 1e-9)
    Wc = (Wc * gain).astype(np.float32)
    bc0 = Y.mean(0) * gain
    core = (pre_all - mu).astype(np.float32) @ Wc          # [B, T, 5] without the class bias
    span = 2.0 * float(np.abs(core).max()) + float(np.abs(bc0).max()) + 1.0
    lo, hi = -span, span                                  # density falls as the blank bias rises; beyond +-span nothing changes
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        b = bc0.copy()
        b[4] += mid
        if greedy_base_count(core + b.astype(np.float32), seq_len).mean() > bases_per_window:
            lo = mid
        else:
            hi = mid
    b = bc0.copy()
    b[4] += 0.5 * (lo + hi)
    out = dict(weights)
    out["rnn_fnn_layer/weights"] = np.ones((2, H), dtype=np.float32)
    out["rnn_fnn_layer/bias"] = (-mu).astype(np.float32)
    out["rnn_fnn_layer/weights_class"] = Wc
    out["rnn_fnn_layer/bias_class"] = b.astype(np.float32)
    return out


def dense_head(spec, w, x_calib, seq_len, bases_per_window):
    """A trained-CTC-like head that EMITS: most frames blank, a base every few frames -- `bases_per_window` on the calibration
    windows (a trained Chiron model decodes 16 .. 45 bases per 400-sample window: chiron/example_data/DNA/output/segments).
    fit_emitting_head on the float64 oracle's recurrent output: the class layer is FITTED, the way a trained head is.  Needs
    cells that follow their input: trained_like_weights(forget_mean=-2)."""
    from oracle import nn_oracle
    sd = spec.to_dict()
    w64 = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    x = np.asarray(x_calib, dtype=np.float64)
    h = nn_oracle.rnn_forward(nn_oracle.cnn_forward(x, sd, w64), seq_len, sd, w64)          # [B, T, 2H]
    return fit_emitting_head(w, h, x, seq_len, bases_per_window, hidden=spec.hidden)
