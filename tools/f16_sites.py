#!/usr/bin/env python3
"""BASELINE configs[4], round-5 review item 3: WHERE do the f16 engine's bases flip?  CPU only (tools/f16_study.py's instrumented
float64 network): the f16 engine's roundings -- weights to halves, every stored activation to halves -- with ONE activation site at a
time kept wide (leave-one-out), and with only one site rounded (one-in), in the two regimes of the review:

  peaked   trained-like weights (tests/regimes.py) under a peaked head: heterogeneous filters, the WEIGHTS' rounding dominates
  density  trained-like weights whose cells follow their input under a head FITTED to emit ~30 bases per window (regimes.dense_head,
           the regime of test_greedy_strings_at_basecalling_density and of bench.py's realistic_density leg): head weights of
           order 100, the ACTIVATIONS' rounding dominates

Sites (f16_study.SITES): conv2a_k / conv2b_k / block_k = the stored activations of residual block k; h_l = the recurrent operand of
LSTM layer l; out_l = layer l's stored output (read by the next projection / the FC head).

    python tools/f16_sites.py [--windows 64] [--regimes peaked,density]  ->  gpurun_out/f16_sites.json
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
import f16_study as fs                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=64)
    ap.add_argument("--regimes", default="peaked,density")
    ap.add_argument("--weights", default="f16", help="f16 | hilo | exact: how the weights are stored in every variant")
    a = ap.parse_args()
    spec = ca.dna_default_spec()
    L, jump = 400, 390
    wq = {"f16": fs.f16, "hilo": fs.hilo, "exact": fs.ident}[a.weights]
    out = {}
    for regime in a.regimes.split(","):
        x, ln = pb.windows(jump * (a.windows - 1) + 200, L, jump, 4711)
        sl = ca.seq_len_for_engine(ln, 1.0)
        if regime == "peaked":
            w = regimes.peaked_head(regimes.trained_like_weights(spec, x[:24], seed=5)[0])
        else:
            w, _ = regimes.trained_like_weights(spec, x[:24], seed=5, forget_mean=-2.0)
            w = regimes.dense_head(spec, w, x[:32], sl[:32], 30)
        ref = fs.forward(x, sl, spec, w, fs.ident, fs.ident)
        rows_ref, _ = ctc_oracle.greedy_decode(ref, sl)
        mask = (np.arange(ref.shape[1])[None, :] < sl[:, None])
        rec = {"windows": int(x.shape[0]), "bases_per_window": sum(len(r) for r in rows_ref) / float(x.shape[0]), "weights": a.weights, "variants": {}}

        def run(name, sites, wq_=wq):
            got = fs.forward(x, sl, spec, w, wq_, sites)
            d = np.abs(got - ref)[mask]
            rows, _ = ctc_oracle.greedy_decode(got, sl)
            same = float(np.mean([list(p) == list(q) for p, q in zip(rows, rows_ref)]))
            rec["variants"][name] = {"identical_windows": same, "logits_mean": float(d.mean()), "logits_p999": float(np.quantile(d, 0.999))}
            print("%-8s %-44s identical %.3f  logits mean %.3g p99.9 %.3g" % (regime, name, same, d.mean(), np.quantile(d, 0.999)), flush=True)

        allf = {s: fs.f16 for s in fs.SITES}
        run("weights only (activations wide)", {})
        run("activations only (weights exact)", allf, fs.ident)
        run("the f16 engine (everything f16)", allf)
        groups = {"cnn": [s for s in fs.SITES if s.startswith(("conv", "block"))], "h (recurrent operands)": ["h_1", "h_2", "h_3"],
                  "out (stored layer outputs)": ["out_1", "out_2", "out_3"], "out_3 (what the FC head reads)": ["out_3"],
                  "block_3 (the features)": ["block_3"], "h_3": ["h_3"], "layer 3 (h_3 + out_3)": ["h_3", "out_3"],
                  "layers 2+3 (h, out)": ["h_2", "out_2", "h_3", "out_3"]}
        for g, members in groups.items():
            run("all f16 EXCEPT " + g, {s: fs.f16 for s in fs.SITES if s not in members})
        for g, members in groups.items():
            run("ONLY %s f16 (weights exact)" % g, {s: fs.f16 for s in members}, fs.ident)
        out[regime] = rec
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "f16_sites_%s.json" % a.weights), "w"), indent=1)


if __name__ == "__main__":
    main()
