#!/usr/bin/env python3
"""What does half precision cost THIS network, and where?  (BASELINE configs[4]; round-3 review, weak point 2.)

CPU only, no kernel involved: the float64 oracle of oracle/nn_oracle.py is re-run with roundings inserted exactly where an
f16 engine has them, one kind at a time, on trained-checkpoint-like weights with a peaked (trained-CTC-like) head:

  weights     conv filters (BN folded in, as the engine stores them) and LSTM kernels rounded to f16
  activations every conv layer's output, the CNN features, every recurrent layer's h rounded to f16 (accumulation, z, gates,
              cell state, logits stay wide, as in the engine)
  both        = what the f16 engine computes (up to accumulation order)

and the two repairs the review proposed:

  unfolded    f16(W) with the BN scale applied in fp32 in the epilogue, instead of f16(W * inv)
  hi+lo       weights as hi + lo half pairs (W to 2^-22: two MFMAs on the 16x faster pipe), activations still f16
  hi+lo both  weights AND activations as hi + lo pairs (= the engine's fp32-split dtype: three MFMAs)

For each: logits deviation from float64 (max, 99.9 %, mean) and the greedy decode (identical windows, edits per window).

    python tools/f16_study.py [--windows 48] [--weight-seeds 5,6]  ->  gpurun_out/f16_study.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import chiron_amd as ca                      # noqa: E402
from oracle import nn_oracle, ctc_oracle     # noqa: E402
import regimes                               # noqa: E402
import parity_budget as pb                   # noqa: E402


def f16(a):
    return np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float64)


def hilo(a):
    hi = f16(a)
    return hi + f16(np.asarray(a, dtype=np.float64) - hi)


def ident(a):
    return np.asarray(a, dtype=np.float64)


SITES = ("conv2a_1", "conv2b_1", "block_1", "conv2a_2", "conv2b_2", "block_2", "conv2a_3", "conv2b_3", "block_3",
         "h_1", "h_2", "h_3", "out_1", "out_2", "out_3")


def forward(x, sl, spec, w, wq, aq, fold=True, wq_lstm=None, bias_correct=False):
    """float64 network with weight rounding wq and activation rounding aq; fold: BN scale inside the rounded filter.
    aq may be a dict {site: rounding} over SITES (a missing site stays wide): conv2a_k / conv2b_k / block_k = the three stored
    activations of residual block k (block_3 = the CNN features), h_l = the recurrent operand of layer l (what the next step's
    h @ W_hh reads), out_l = the layer's STORED output (what the next layer's projection, or the FC head, reads)."""
    sd = spec.to_dict()
    if isinstance(aq, dict):
        sites = aq
        aqs = lambda name: sites.get(name, ident)
    else:
        one = aq
        aqs = lambda name: one
    w = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}

    def conv(xx, site, stride, bn, relu):
        f = w[site + "/weights"]
        f = f.reshape(f.shape[-3], f.shape[-2], f.shape[-1])
        if bn:
            inv = (1.0 / np.sqrt(w[site + "_bn/pop_var"] + nn_oracle.BN_EPS)) * w[site + "_bn/scale"]
            shift = w[site + "_bn/offset"] - w[site + "_bn/pop_mean"] * inv
            if bias_correct:      # best case of a bias correction: the input channels' true means on this data
                dW = wq(f * inv) - f * inv
                shift = shift - np.einsum("k,tkn->n", xx.mean(axis=(0, 1)), dW)
            y = nn_oracle.conv1d_same(xx, wq(f * inv), stride) + shift if fold else nn_oracle.conv1d_same(xx, wq(f), stride) * inv + shift
        else:
            y = nn_oracle.conv1d_same(xx, wq(f), stride)
            if bias_correct:
                y = y - np.einsum("k,tkn->n", xx.mean(axis=(0, 1)), wq(f) - f)
        return np.maximum(y, 0) if relu else y

    a = np.asarray(x, dtype=np.float64)[:, :, None]        # the raw signal: integers below 2048 are exact halves
    for bi, blk in enumerate(sd["cnn"], 1):
        n, s = blk["name"], blk.get("stride", 1)
        b1 = conv(a, n + "/branch1/conv1", s, blk["i_bn"], False)
        c = aqs("conv2a_%d" % bi)(conv(a, n + "/branch2/conv2a", 1, True, True))
        c = aqs("conv2b_%d" % bi)(conv(c, n + "/branch2/conv2b", s, True, True))
        c = conv(c, n + "/branch2/conv2c", 1, True, False)
        a = aqs("block_%d" % bi)(np.maximum(b1 + c, 0))      # the engine adds branch1 inside the conv2c accumulator
    H = spec.hidden
    wl = dict(w)
    for k in w:
        if k.endswith("lstm_cell/kernel"):
            q = wq_lstm or wq
            if isinstance(q, tuple):       # (x-part rounding, h-part rounding): the kernel's rows are [x ; h]
                nin = w[k].shape[0] - H
                wl[k] = np.concatenate([q[0](w[k][:nin]), q[1](w[k][nin:])], axis=0)
            else:
                wl[k] = q(w[k])
    prev = a
    for layer in range(spec.rnn_layers):
        # h is rounded where it is STORED: the stage's output (read by the next layer and by the recurrence itself)
        prev = rnn_layer_q(prev, sl, sd, wl, layer, (aqs("h_%d" % (layer + 1)), aqs("out_%d" % (layer + 1))), w if bias_correct == 2 else None)
    return nn_oracle.fc_head(prev, wl)


def rnn_layer_q(x, sl, sd, w, layer, aq, w_exact=None):
    """nn_oracle.rnn_layer_forward with h rounded by aq at every step (the recurrent input is the stored, rounded h)"""
    r = sd["rnn"]
    H = r["hidden"]
    outs = []
    for di, (d, rev) in enumerate((("fw", False), ("bw", True))):
        if r["kind"] == "stack":
            p = "BDLSTM_rnn/cell_%d/bidirectional_rnn/%s/lstm_cell/" % (layer, d)
            xin = x
        else:
            p = "BDGRU_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell/" % (d, layer)
            xin = x if layer == 0 else x[:, :, di * H:(di + 1) * H]
        bias = w[p + "bias"]
        if w_exact is not None:        # best-case bias correction: the inputs' true means (x rows: this layer's input over the valid
            # frames; h rows: this direction's own exact output, one pass with the uncorrected bias first)
            valid = (np.arange(x.shape[1])[None, :] < np.asarray(sl)[:, None])
            dW = w[p + "kernel"] - w_exact[p + "kernel"]
            nin = dW.shape[0] - H
            mx = xin[valid].mean(axis=0)
            h0 = lstm_direction_q(np.ascontiguousarray(xin), sl, w[p + "kernel"], bias, rev, aq)
            mh = h0[valid].mean(axis=0)
            bias = bias - mx @ dW[:nin] - mh @ dW[nin:]
        outs.append(lstm_direction_q(np.ascontiguousarray(xin), sl, w[p + "kernel"], bias, rev, aq))
    return np.concatenate(outs, axis=2)


def lstm_direction_q(x, seq_len, kernel, bias, reverse, aq):
    aq_rec, aq_out = aq if isinstance(aq, tuple) else (aq, aq)
    B, T, _ = x.shape
    H = kernel.shape[1] // 4
    out = np.zeros((B, T, H))
    h = np.zeros((B, H))
    c = np.zeros((B, H))
    seq_len = np.asarray(seq_len).astype(np.int64)
    rows = np.arange(B)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for step in range(T):
        active = step < seq_len
        if not active.any():
            break
        t_idx = np.where(active, seq_len - 1 - step, 0) if reverse else np.full(B, step)
        z = np.concatenate([x[rows, t_idx], h], axis=1) @ kernel + bias
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c_new = sig(f + 1.0) * c + sig(i) * np.tanh(j)
        h_wide = sig(o) * np.tanh(c_new)
        h_new = aq_rec(h_wide)
        m = active[:, None]
        c = np.where(m, c_new, c)
        h = np.where(m, h_new, h)
        out[rows[active], t_idx[active]] = aq_out(h_wide)[active]
    return out


def lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca_ in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca_ != cb)))
        prev = cur
    return prev[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=48)
    ap.add_argument("--weight-seeds", default="5,6")
    ap.add_argument("--topology", default="dna")
    a = ap.parse_args()
    spec = ca.dna_default_spec() if a.topology == "dna" else ca.rna_default_spec()
    L, jump = (400, 390) if a.topology == "dna" else (500, 490)
    variants = [("weights f16", f16, ident, True), ("activations f16", ident, f16, True), ("both f16 (the f16 engine)", f16, f16, True),
                ("both f16, BN scale in the epilogue", f16, f16, False), ("weights hi+lo, activations f16", hilo, f16, True),
                ("weights f16, activations hi+lo", f16, hilo, True), ("both hi+lo (fp32-split dtype)", hilo, hilo, True),
                ("conv weights hi+lo, LSTM kernels f16, activations f16", hilo, f16, True, f16),
                ("conv weights f16, LSTM kernels hi+lo, activations f16", f16, f16, True, hilo),
                ("both f16 + conv bias correction with the data's own channel means", f16, f16, True, None, True),
                ("conv f16 + bias correction, LSTM kernels hi+lo", f16, f16, True, hilo, True),
                ("all weights f16 + bias correction of conv AND LSTM (inputs' true means)", f16, f16, True, None, 2),
                ("conv + LSTM x-part hi+lo, W_hh f16, activations f16", hilo, f16, True, (hilo, f16)),
                ("conv + W_hh hi+lo, LSTM x-part f16, activations f16", hilo, f16, True, (f16, hilo))]
    out = []
    for k, ws in enumerate(int(v) for v in a.weight_seeds.split(",")):
        x, ln = pb.windows(jump * (a.windows - 1) + 200, L, jump, 67 + 10 * k)
        w, _ = regimes.trained_like_weights(spec, x[:24], seed=ws)
        w = regimes.peaked_head(w)
        sl = ca.seq_len_for_engine(ln, L / nn_oracle.output_len(L, spec.to_dict()))
        ref = forward(x, sl, spec, w, ident, ident)
        chk, _ = nn_oracle.inference(x, sl, spec.to_dict(), w, dtype=np.float64)
        assert np.abs(ref - chk).max() < 1e-9, np.abs(ref - chk).max()       # the instrumented network IS the oracle
        rows_ref, _ = ctc_oracle.greedy_decode(ref, sl)
        mask = (np.arange(ref.shape[1])[None, :] < sl[:, None])
        srt = np.sort(ref, axis=-1)
        margin = (srt[..., -1] - srt[..., -2])[mask]
        rec = {"topology": a.topology, "weight_seed": ws, "windows": int(x.shape[0]), "bases": int(sum(len(r) for r in rows_ref)),
               "margin_median": float(np.median(margin)), "frames_with_margin_above_1.2": float((margin > 1.2).mean()), "variants": {}}
        for name, wq, aq, fold, *rest in variants:
            got = forward(x, sl, spec, w, wq, aq, fold, rest[0] if rest else None, bias_correct=len(rest) > 1 and rest[1])
            d = np.abs(got - ref)[mask]
            rows, _ = ctc_oracle.greedy_decode(got, sl)
            dist = [0 if list(p) == list(q) else lev(list(p), list(q)) for p, q in zip(rows, rows_ref)]
            rec["variants"][name] = {"logits_max": float(d.max()), "logits_p999": float(np.quantile(d, 0.999)), "logits_mean": float(d.mean()),
                                     "identical_windows": float(np.mean([v == 0 for v in dist])), "edits_per_window": float(np.mean(dist))}
            print("%s seed %d  %-40s logits max %.3g p99.9 %.3g mean %.3g | identical %.3f edits/window %.3f" % (
                a.topology, ws, name, d.max(), np.quantile(d, 0.999), d.mean(), np.mean([v == 0 for v in dist]), np.mean(dist)))
            sys.stdout.flush()
        out.append(rec)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "f16_study_%s.json" % a.topology), "w"), indent=1)


if __name__ == "__main__":
    main()
