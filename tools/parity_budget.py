#!/usr/bin/env python
"""Per-stage error budget of the fp32 engine on trained-checkpoint-like weights (tests/regimes.py), DNA and RNA.

Stages: getcnnfeature (cnn.py:334-371) -> LSTM layer 1, 2, 3 (rnn.py:20-97 / :99-174) -> FC head logits (rnn.py:72-96).
For every stage s three numbers per implementation (the HIP engine, and the float32 numpy restatement of oracle/nn_oracle.py):

  total   | impl_s - float64_s |            what has accumulated up to and including the stage
  local   | impl_s - F64_s(impl_{s-1}) |    the error BORN in the stage: the float64 oracle's stage applied to the
                                            implementation's own previous-stage output
  (max and rms over the valid frames)

and, for the logits, the greedy-decode comparison the north star asks for (chiron_eval.py:485-487): per window the engine's
greedy string against the float64 oracle's, every frame whose argmax differs with the float64 top-1 / top-2 margin.

The engine's per-layer outputs come from chiron_engine_rnn_output on engines built with 1, 2 and 3 rnn_layers (the model
descriptor is data-driven; the head is not used for those).  Run once per gate-math build (CHIRON_AMD_LIB=build/libchiron_lstm_
CHIRON_GATE_MATH_<n>.so; tools/variants.sh --product lstm CHIRON_GATE_MATH 1 2) -- the tag names the output file.

  python tools/parity_budget.py [tag] [--windows 24] [--peaked]  ->  gpurun_out/parity_budget_<tag>.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import chiron_amd as ca                      # noqa: E402
from oracle import nn_oracle, ctc_oracle     # noqa: E402  (checker only: this is a measurement tool, not a product path)
import regimes                               # noqa: E402


def windows(n_samples, L, jump, seed):
    from chiron_amd import signal_io
    sig = ca.synthetic_signal(1, n_samples, seed=seed)[0]
    ev, ln = signal_io.window_signal(sig, 0, jump, L)
    return np.array(ev, dtype=np.float32), np.asarray(ln, dtype=np.int64)      # a writable copy: window_signal's rows are read-only views


def stats(a, b, mask=None):
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    if mask is not None:
        d = d[mask]
    return {"max": float(d.max()), "rms": float(np.sqrt((d ** 2).mean()))}


def spec_with_layers(spec, n):
    return ca.ModelSpec(spec.blocks, spec.rnn_kind, n, spec.hidden, spec.classes, spec.bn_mode, spec.stem)


def engine_stages(spec, w, x, ln, L):
    """-> sl, {features, lstm1.., logits} of the HIP engine (default forms)"""
    out = {}
    nl = spec.rnn_layers
    for n in range(1, nl + 1):
        with ca.Engine(spec_with_layers(spec, n), w, max_batch=x.shape[0], segment_len=L) as eng:
            sl = ca.seq_len_for_engine(ln, eng.ratio)
            res = eng.infer(x, sl, want_logits=True)
            out["lstm%d" % n] = eng.rnn_output()
            if n == nl:
                out["features"] = eng.features()
                out["logits"] = res.logits
                out["decoded"] = res.decoded
    return sl, out


def oracle_stages(spec, w, x, sl, dtype):
    sd = spec.to_dict()
    ww = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    out = {"features": nn_oracle.cnn_forward(np.asarray(x, dtype=dtype), sd, ww)}
    prev = out["features"]
    for n in range(1, spec.rnn_layers + 1):
        prev = nn_oracle.rnn_layer_forward(prev, sl, sd, ww, n - 1)
        out["lstm%d" % n] = prev
    out["logits"] = nn_oracle.fc_head(prev, ww)
    return out


F32 = np.float32


# ---- float32 restatements of getcnnfeature in OTHER summation orders than oracle/nn_oracle.py's.  The logits' deviation is the
# features' rounding error amplified by the recurrent stack, and the amplification is ill-conditioned (one window of a weight set
# can carry 1000 x: tools/cnn_error_structure.py) -- so "engine / numpy-fp32" against ONE realisation of float32 rounding is a noisy
# statistic.  These are the other realisations an fp32 implementation of the SAME formulas may legitimately have:
#   natural  nn_oracle in float32: BLAS, one K = C chain per tap, BN applied to the rounded sum
#   folded   BN folded into the filters (what the engine and any inference runtime does), BLAS chains
#   chain    folded, ONE sequential fmaf chain over the whole K starting from the shift: what an MFMA accumulator (or a plain loop) does
def fold(w, site):
    """engine.hip:fold_bn in float32: inv = (1/sqrt(var + eps))*scale, shift = offset - mean*inv"""
    sc, of, mu, var = [w[site + "_bn/" + k].astype(F32) for k in ("scale", "offset", "pop_mean", "pop_var")]
    inv = ((F32(1.0) / np.sqrt(var + F32(nn_oracle.BN_EPS))).astype(F32) * sc).astype(F32)
    return inv, (of - (mu * inv).astype(F32)).astype(F32)


def fma32(a, b, c):
    """fl32(a*b + c) with one rounding: the product of two floats is exact in float64; the float64 sum is then rounded to float32
    (double rounding differs from a true fmaf in about one case in 2^29: immaterial for error statistics)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def conv_chain(x, wf, shift, stride):
    """ONE sequential fmaf chain per output over K = taps x channels, accumulator initialised with the shift (gemm.hip: the shift is
    the C operand of a tile's first MFMA; K order = tap-major, channels ascending, as upload_gemm lays Wt out)"""
    B, W, cin = x.shape
    k, _, cout = wf.shape
    out, left, right = nn_oracle.same_padding(W, k, stride)
    xp = np.zeros((B, W + left + right, cin), dtype=F32)
    xp[:, left:left + W] = x
    acc = np.broadcast_to(shift.astype(F32), (B, out, cout)).copy()
    for tap in range(k):
        xs = xp[:, tap:tap + (out - 1) * stride + 1:stride]
        for c in range(cin):
            acc = fma32(xs[:, :, c:c + 1], wf[tap, c][None, None, :], acc)
    return acc


def block_f32(x, w, blk, order):
    """one residual block (cnn.py:234-262) in float32, population BN, in summation order `order`"""
    if order == "natural":
        return nn_oracle.residual_layer(x.astype(F32), {k: v.astype(F32) for k, v in w.items()}, blk, "population")
    n, s = blk["name"], blk.get("stride", 1)

    def conv(xx, site, stride, bn, relu):
        W = w[site + "/weights"]
        W = W.reshape(W.shape[-3], W.shape[-2], W.shape[-1]).astype(F32)
        if bn:
            inv, sh = fold(w, site)
            W = (W * inv[None, None, :]).astype(F32)
        else:
            sh = np.zeros(W.shape[-1], F32)
        if order == "chain" and W.shape[1] > 1:
            y = conv_chain(xx.astype(F32), W, sh, stride)
        else:
            y = (nn_oracle.conv1d_same(xx.astype(F32), W, stride) + sh).astype(F32)
        return np.maximum(y, 0) if relu else y

    b1 = conv(x, n + "/branch1/conv1", s, blk["i_bn"], False)
    a = conv(x, n + "/branch2/conv2a", 1, True, True)
    b = conv(a, n + "/branch2/conv2b", s, True, True)
    c = conv(b, n + "/branch2/conv2c", 1, True, False)
    return np.maximum((b1 + c).astype(F32), 0)


def restatement_blocks(x, spec_d, w, order):
    out, p = [], np.asarray(x, dtype=F32)[:, :, None]
    for blk in spec_d["cnn"]:
        p = block_f32(p, w, blk, order)
        out.append(p)
    return out


def local_reference(spec, w, stage, prev_impl, x, sl):
    """float64 oracle's `stage` applied to an implementation's previous-stage output"""
    sd = spec.to_dict()
    ww = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    if stage == "features":
        return nn_oracle.cnn_forward(np.asarray(x, dtype=np.float64), sd, ww)       # the input is exact: local == total
    p = np.asarray(prev_impl, dtype=np.float64)
    if stage.startswith("lstm"):
        return nn_oracle.rnn_layer_forward(p, sl, sd, ww, int(stage[4:]) - 1)
    return nn_oracle.fc_head(p, ww)


def propagate_to_logits(spec, w, stage, impl_out, sl):
    """the float64 oracle's REMAINING stages applied to an implementation's output of `stage`: the logits the network would give if
    everything after that stage were exact -- their distance from the float64 logits is what the stages up to `stage` are
    responsible for at the logits (errors born later are excluded; the network's own amplification is included)."""
    sd = spec.to_dict()
    ww = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    p = np.asarray(impl_out, dtype=np.float64)
    if stage == "logits":
        return p
    first = 0 if stage == "features" else int(stage[4:])
    for n in range(first, spec.rnn_layers):
        p = nn_oracle.rnn_layer_forward(p, sl, sd, ww, n)
    return nn_oracle.fc_head(p, ww)


def greedy_report(logits_impl, logits64, sl, err_max):
    """chiron_eval.py:485-487 on both logits: identical windows, and every frame whose argmax differs with its float64 margin"""
    B = logits64.shape[0]
    rows_i, _ = ctc_oracle.greedy_decode(np.asarray(logits_impl, dtype=np.float32), sl)
    rows_o, _ = ctc_oracle.greedy_decode(np.asarray(logits64, dtype=np.float64), sl)
    same = [list(a) == list(b) for a, b in zip(rows_i, rows_o)]
    T = logits64.shape[1]
    mask = np.arange(T)[None, :] < np.asarray(sl)[:, None]
    srt = np.sort(logits64, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    flip = (np.argmax(logits_impl, axis=-1) != np.argmax(logits64, axis=-1)) & mask
    flips = [{"window": int(b), "frame": int(t), "float64_margin": float(margin[b, t])} for b, t in zip(*np.nonzero(flip))]
    m = margin[mask]
    return {"windows": B, "identical_windows": int(sum(same)), "identical_fraction": float(np.mean(same)),
            "bases_float64": int(sum(len(r) for r in rows_o)),
            "frames": int(mask.sum()), "flipped_frames": len(flips), "flips": flips[:200],
            "largest_margin_of_a_flipped_frame": max([f["float64_margin"] for f in flips], default=0.0),
            "logit_error_max": err_max,
            "frames_with_margin_below_twice_the_logit_error": int((m < 2 * err_max).sum()),
            "smallest_margin": float(m.min()), "margin_quantiles": {q: float(np.quantile(m, float(q))) for q in ("0.001", "0.01", "0.1", "0.5")}}


def fp32_order_logits(spec, w, x, sl, order):
    """logits (and features) of a float32 pipeline whose CNN sums in `order` ("folded" / "chain"; "natural" is oracle_stages with
    float32); the recurrent stack and the head are nn_oracle in float32 in every order"""
    sd = spec.to_dict()
    w32 = {k: np.asarray(v, dtype=F32) for k, v in w.items()}
    fea = restatement_blocks(x, sd, w, order)[-1]
    p = fea
    for n in range(spec.rnn_layers):
        p = nn_oracle.rnn_layer_forward(p, sl, sd, w32, n)
    return fea, nn_oracle.fc_head(p, w32)


def budget(topology, n_windows, peaked, seed=67, weight_seed=5, orders=()):
    spec = ca.dna_default_spec() if topology == "dna" else ca.rna_default_spec()
    L, jump = (400, 390) if topology == "dna" else (500, 490)
    x, ln = windows(jump * (n_windows - 1) + 200, L, jump, seed)
    ln = ln.copy()
    ln[2], ln[5] = L // 3, 0
    w, _ = regimes.trained_like_weights(spec, x[:24], seed=weight_seed)
    if peaked:
        w = regimes.peaked_head(w)
    t0 = time.time()
    sl, eng = engine_stages(spec, w, x, ln, L)
    o64 = oracle_stages(spec, w, x, sl, np.float64)
    n32 = oracle_stages(spec, w, x, sl, np.float32)
    T = o64["logits"].shape[1]
    fmask = (np.arange(T)[None, :] < np.asarray(sl)[:, None])[..., None]
    order = ["features"] + ["lstm%d" % n for n in range(1, spec.rnn_layers + 1)] + ["logits"]
    rep = {"topology": topology, "windows": int(x.shape[0]), "peaked_head": bool(peaked), "signal_seed": seed, "weight_seed": weight_seed, "stages": {}}
    for impl_name, impl in (("engine", eng), ("numpy_fp32", n32)):
        prev = None
        for s in order:
            full = np.broadcast_to(fmask, o64[s].shape) if s != "features" else None
            e = rep["stages"].setdefault(s, {"scale_rms": float(np.sqrt((o64[s] ** 2).mean())), "scale_max": float(np.abs(o64[s]).max())})
            e[impl_name + "_total"] = stats(impl[s], o64[s], full)
            loc = local_reference(spec, w, s, prev, x, sl)
            e[impl_name + "_local"] = stats(impl[s], loc, full)
            # what the stages up to and including s cost AT THE LOGITS (float64 oracle from here on)
            e[impl_name + "_at_logits"] = stats(propagate_to_logits(spec, w, s, impl[s], sl), o64["logits"], np.broadcast_to(fmask, o64["logits"].shape))
            prev = impl[s]
    lg = rep["stages"]["logits"]
    rep["logits_ratio_engine_over_numpy_fp32"] = {k: lg["engine_total"][k] / max(lg["numpy_fp32_total"][k], 1e-30) for k in ("max", "rms")}
    # other float32 summation orders of the same formulas: the spread of the reference itself (see fp32_order_logits)
    lmask = np.broadcast_to(fmask, o64["logits"].shape)
    rep["numpy_fp32_orders"] = {"natural": {"features": stats(n32["features"], o64["features"]), "logits": lg["numpy_fp32_total"]}}
    for o in orders:
        fea_o, lg_o = fp32_order_logits(spec, w, x, sl, o)
        rep["numpy_fp32_orders"][o] = {"features": stats(fea_o, o64["features"]), "logits": stats(lg_o, o64["logits"], lmask)}
    rep["greedy_engine_vs_float64"] = greedy_report(eng["logits"], o64["logits"], sl, lg["engine_total"]["max"])
    rep["greedy_numpy_fp32_vs_float64"] = greedy_report(n32["logits"], o64["logits"], sl, lg["numpy_fp32_total"]["max"])
    # the device's own decode of its own logits is the oracle's decode of those logits (bit-exact integer work)
    rows, _ = ctc_oracle.greedy_decode(eng["logits"], sl)
    idx, val, shape = ctc_oracle.rows_to_sparse(rows, x.shape[0])
    rep["device_decode_equals_oracle_decode_of_device_logits"] = bool(
        np.array_equal(idx, eng["decoded"].indices) and np.array_equal(val, eng["decoded"].values) and np.array_equal(shape, eng["decoded"].dense_shape))
    rep["seconds"] = time.time() - t0
    return rep


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("tag", nargs="?", default="default")
    ap.add_argument("--windows", type=int, default=24)
    ap.add_argument("--peaked", action="store_true")
    ap.add_argument("--weight-seeds", default="5", help="comma separated: one budget per topology and seed (the max-norm ratio of two "
                    "amplified rounding errors is a noisy statistic; several weight sets show its spread)")
    ap.add_argument("--orders", default="", help="comma separated float32 summation orders besides nn_oracle's: folded, chain")
    a = ap.parse_args()
    tag, n, peaked = a.tag, a.windows, a.peaked
    seeds = [int(v) for v in a.weight_seeds.split(",")]
    out = {"tag": tag, "lib": os.environ.get("CHIRON_AMD_LIB") or "product",
           "budgets": [budget(t, n, peaked, seed=67 + 10 * k, weight_seed=ws, orders=tuple(v for v in a.orders.split(",") if v))
                       for t in ("dna", "rna") for k, ws in enumerate(seeds)]}
    if len(seeds) > 1:
        out["ratio_engine_over_numpy_fp32_by_seed"] = {
            t: {m: [b["logits_ratio_engine_over_numpy_fp32"][m] for b in out["budgets"] if b["topology"] == t] for m in ("max", "rms")}
            for t in ("dna", "rna")}
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "parity_budget_%s%s.json" % (tag, "_peaked" if peaked else ""))
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for b in out["budgets"]:
        print("== %s (%s%s, weights %d): logits engine/numpy-fp32 max %.2f rms %.2f" % (b["topology"], tag, " peaked" if peaked else "", b["weight_seed"],
              b["logits_ratio_engine_over_numpy_fp32"]["max"], b["logits_ratio_engine_over_numpy_fp32"]["rms"]))
        for s, e in b["stages"].items():
            print("  %-9s scale %8.3g | engine total %.3g (rms %.3g) local %.3g (rms %.3g) at-logits %.3g (rms %.3g) | numpy-fp32 total %.3g (rms %.3g) "
                  "local %.3g (rms %.3g) at-logits %.3g (rms %.3g)" % (
                      s, e["scale_rms"], e["engine_total"]["max"], e["engine_total"]["rms"], e["engine_local"]["max"], e["engine_local"]["rms"],
                      e["engine_at_logits"]["max"], e["engine_at_logits"]["rms"],
                      e["numpy_fp32_total"]["max"], e["numpy_fp32_total"]["rms"], e["numpy_fp32_local"]["max"], e["numpy_fp32_local"]["rms"],
                      e["numpy_fp32_at_logits"]["max"], e["numpy_fp32_at_logits"]["rms"]))
        print("  float32 summation orders at the logits (max / rms): " + "  ".join("%s %.3g / %.3g" % (o, v["logits"]["max"], v["logits"]["rms"])
                                                                                   for o, v in b["numpy_fp32_orders"].items()))
        g = b["greedy_engine_vs_float64"]
        print("  greedy: %d / %d windows identical, %d flipped frames (largest float64 margin %.3g, logit error %.3g)" % (
            g["identical_windows"], g["windows"], g["flipped_frames"], g["largest_margin_of_a_flipped_frame"], g["logit_error_max"]))


if __name__ == "__main__":
    main()
