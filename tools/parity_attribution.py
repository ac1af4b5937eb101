#!/usr/bin/env python
"""Where one weight set's logit error comes from: per residual block, per convolution form (round-5 review, item 1c).

The case the review names: DNA, trained-like weight set 8, 24 windows (tests/regimes.py; the pair of
profiles/r05_parity_budget_dna_8.json): the engine's logits are 5.2e-4 from float64, the float32 restatement that sums like an MFMA
accumulator (`chain`) 7.6e-5, the one with BN folded and BLAS sums (`folded`) 5.07e-4.  Why does the engine read like `folded`?

For every implementation -- the engine in each convolution form (INTEGRATION.md switches) and the float32 restatements `natural`,
`folded`, `chain` of tools/parity_budget.py -- and every residual block k (cnn.py:234-262):

  born_k      d_k = block_k(impl) - Block64_k(block_{k-1}(impl)): the error BORN in block k (float64 block on the implementation's own input)
  at_logits_k the float64 network from block k + 1 on, fed Block64-exact activations + d_k ALONE: what block k's rounding costs at the logits
  signed_k    the same at ONE logit -- the (window, frame, class) where the engine's error is largest -- with its sign, so that
              signed_1 + signed_2 + signed_3 (+ the recurrent stack's own rounding) adds up to the implementation's error there

and for the worst window: `draws` float32 realisations of the whole pipeline (tools/parity_dist.py:realisation) -- the distribution that
window's error is a draw from -- with the percentile of each implementation in it.

  python tools/parity_attribution.py [--topology dna --weight-seed 8 --windows 24 --draws 48]  ->  gpurun_out/parity_attribution_<topology>_<seed>.json
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
import chiron_amd as ca            # noqa: E402
from oracle import nn_oracle       # noqa: E402  (checker only: measurement tool)
import regimes                     # noqa: E402
import parity_budget as pb         # noqa: E402
import parity_dist as pdist        # noqa: E402
import cnn_error_structure as ces  # noqa: E402

FORMS = ces.FORMS + (("winograd-F2", {"CHIRON_WINOGRAD_F2": "1"}),)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="dna")
    ap.add_argument("--weight-seed", type=int, default=8)
    ap.add_argument("--windows", type=int, default=24)
    ap.add_argument("--draws", type=int, default=48)
    ap.add_argument("--no-engine", action="store_true")
    a = ap.parse_args()
    topology, ws, n = a.topology, a.weight_seed, a.windows
    spec = ca.dna_default_spec() if topology == "dna" else ca.rna_default_spec()
    L, jump = (400, 390) if topology == "dna" else (500, 490)
    x, ln = pb.windows(jump * (n - 1) + 200, L, jump, 67 + 10 * (ws - 5))
    ln = ln.copy()
    ln[2], ln[5] = L // 3, 0
    w, _ = regimes.trained_like_weights(spec, x[:24], seed=ws)
    sd = spec.to_dict()
    w64 = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    blocks64, p = [], x.astype(np.float64)[:, :, None]
    for blk in sd["cnn"]:
        p = nn_oracle.residual_layer(p, w64, blk, "population")
        blocks64.append(p)
    T = blocks64[-1].shape[1]
    sl = ca.seq_len_for_engine(ln, L / float(T))
    logits64 = pb.propagate_to_logits(spec, w, "features", blocks64[-1], sl)
    fmask = (np.arange(T)[None, :] < np.asarray(sl)[:, None])[..., None]

    def logits_from_block(k, act):
        """float64 network behind block k (0-based) applied to `act`"""
        q = np.asarray(act, dtype=np.float64)
        for blk in sd["cnn"][k + 1:]:
            q = nn_oracle.residual_layer(q, w64, blk, "population")
        return pb.propagate_to_logits(spec, w, "features", q, sl)

    impls = {"numpy_fp32_" + o: pb.restatement_blocks(x, sd, w, o) for o in ("natural", "folded", "chain")}
    logits_of = {}
    w32 = {k: np.asarray(v, dtype=np.float32) for k, v in w.items()}
    for name, bl in impls.items():
        q = bl[-1]
        for layer in range(spec.rnn_layers):
            q = nn_oracle.rnn_layer_forward(q, sl, sd, w32, layer)
        logits_of[name] = nn_oracle.fc_head(q, w32)
    if not a.no_engine:
        for name, env in FORMS:
            for s in ces.SWITCHES:
                os.environ.pop(s, None)
            os.environ.update(env)
            impls["engine_" + name] = ces.engine_blocks(spec, w, x, ln, L)
            with ca.Engine(spec, w, max_batch=n, segment_len=L) as eng:
                logits_of["engine_" + name] = eng.infer(x, sl, want_logits=True).logits
        for s in ces.SWITCHES:
            os.environ.pop(s, None)
    lead = "engine_default" if not a.no_engine else "numpy_fp32_folded"
    err = (logits_of[lead].astype(np.float64) - logits64) * fmask
    W, t, c = np.unravel_index(np.abs(err).argmax(), err.shape)
    rep = {"topology": topology, "weight_seed": ws, "windows": n, "worst_logit_of": lead, "worst_window": int(W), "worst_frame": int(t), "worst_class": int(c),
           "implementations": {}}
    print("%s weights %d: %s is worst at window %d frame %d class %d: %.3g" % (topology, ws, lead, W, t, c, err[W, t, c]))
    for name, bl in impls.items():
        e = (logits_of[name].astype(np.float64) - logits64) * fmask
        r = {"logits_max": float(np.abs(e).max()), "logits_max_window": int(np.abs(e).max(axis=(1, 2)).argmax()),
             "logits_max_in_worst_window": float(np.abs(e[W]).max()), "signed_error_at_worst_logit": float(e[W, t, c]), "blocks": []}
        prev = x.astype(np.float64)[:, :, None]
        total_signed = 0.0
        for k, blk in enumerate(sd["cnn"]):
            born = bl[k].astype(np.float64) - nn_oracle.residual_layer(prev, w64, blk, "population")
            d = (logits_from_block(k, blocks64[k] + born) - logits64) * fmask
            r["blocks"].append({"born_rms": ces.rms(born), "born_max": float(np.abs(born).max()), "born_rms_in_worst_window": ces.rms(born[W]),
                                "at_logits_max": float(np.abs(d).max()), "at_logits_max_in_worst_window": float(np.abs(d[W]).max()),
                                "signed_at_worst_logit": float(d[W, t, c])})
            total_signed += float(d[W, t, c])
            prev = bl[k].astype(np.float64)
        # what the convolutions are responsible for there, against what the float32 recurrent stack + head add on top
        feat = (logits_from_block(len(sd["cnn"]) - 1, bl[-1]) - logits64) * fmask
        r["features_alone_signed_at_worst_logit"] = float(feat[W, t, c])
        r["sum_of_blocks_signed"] = total_signed
        r["recurrent_stack_and_head_signed"] = float(e[W, t, c] - feat[W, t, c])
        rep["implementations"][name] = r
        print("%-26s logits max %.3g (window %d; in window %d: %.3g) | at the worst logit %+.3g = blocks %s (sum %+.3g; features alone %+.3g) + stack %+.3g | born rms %s" % (
            name, r["logits_max"], r["logits_max_window"], W, r["logits_max_in_worst_window"], r["signed_error_at_worst_logit"],
            " ".join("%+.3g" % b["signed_at_worst_logit"] for b in r["blocks"]), total_signed, r["features_alone_signed_at_worst_logit"],
            r["recurrent_stack_and_head_signed"], " ".join("%.3g" % b["born_rms"] for b in r["blocks"])), flush=True)
    # the distribution that window's error is a draw from
    draws_max, draws_signed = [], []
    for r_ in range(a.draws):
        rng = np.random.RandomState(424243 + r_)
        _, lg = pdist.realisation(spec, w, x, sl, rng)
        e = (lg["plain"].astype(np.float64) - logits64) * fmask
        draws_max.append(float(np.abs(e[W]).max()))
        draws_signed.append(float(e[W, t, c]))
    dm = np.sort(draws_max)
    rep["realisations_in_worst_window"] = {"draws": a.draws, "max_quantiles": {q: float(np.quantile(dm, float(q))) for q in ("0", "0.1", "0.25", "0.5", "0.75", "0.9", "1")},
                                           "signed_at_worst_logit_std": float(np.std(draws_signed)), "signed_at_worst_logit_mean": float(np.mean(draws_signed))}
    for name, r in rep["implementations"].items():
        r["percentile_among_realisations_in_worst_window"] = float((dm < r["logits_max_in_worst_window"]).mean())
    print("window %d over %d float32 realisations: max error quantiles %s; signed error at the worst logit: mean %+.3g std %.3g" % (
        W, a.draws, " ".join("%s: %.3g" % kv for kv in rep["realisations_in_worst_window"]["max_quantiles"].items()),
        rep["realisations_in_worst_window"]["signed_at_worst_logit_mean"], rep["realisations_in_worst_window"]["signed_at_worst_logit_std"]))
    print("percentiles there: " + "  ".join("%s %.2f" % (k, v["percentile_among_realisations_in_worst_window"]) for k, v in rep["implementations"].items()))
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(rep, open(os.path.join(d, "parity_attribution_%s_%d.json" % (topology, ws)), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
