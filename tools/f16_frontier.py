#!/usr/bin/env python3
"""BASELINE configs[4] ("CDNA4 fp16 MFMA path, tolerance check vs fp32"), round-5 review item 3: the measured frontier
time per 4096 windows  x  windows whose greedy string equals the fp32 engine's, for every half-precision mode the engine has, in
the regimes the review names:

  peaked          trained-like weights (tests/regimes.py: heterogeneous filters, calibrated BN) under a peaked head  -- weights' rounding dominates
  density         the headline's synthetic weights, cells that follow their input, a head FITTED to emit 43.9 bases per window (bench.py's
                  realistic_density leg)                                                                              -- activations' rounding dominates
  peaked-density  trained-like weights whose cells follow their input under a fitted head (test_greedy_strings_at_basecalling_density) -- both

modes: fp16 (halves everywhere) without / with its bias correction, each with the last layer's output to the FC head as halves
(CHIRON_F16_LASTH16=1: rounds 2 .. 5) or as fp32 (round 6); fp16-w2 (exact weights) the same two ways; fp32-split.

    python tools/f16_frontier.py [--windows 1100]  ->  gpurun_out/f16_frontier.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import chiron_amd as ca                      # noqa: E402
import regimes                               # noqa: E402
import parity_budget as pb                   # noqa: E402

L, JUMP = 400, 390
MODES = [("fp16, lasth f16", "fp16", False, True), ("fp16, lasth f32", "fp16", False, False),
         ("fp16 + bias correction, lasth f16", "fp16", True, True), ("fp16 + bias correction, lasth f32", "fp16", True, False),
         ("fp16-w2, lasth f16", "fp16-w2", False, True), ("fp16-w2, lasth f32", "fp16-w2", False, False),
         ("fp32-split", "fp32-split", False, False)]


def rows_of(res, n):
    return np.split(res.decoded.values, np.cumsum(np.bincount(res.decoded.indices[:, 0], minlength=n))[:-1])


def time_4096(spec, w, x, sl, dtype, calibrate):
    B16 = 4096
    reps = -(-B16 // x.shape[0])
    xx = np.ascontiguousarray(np.concatenate([x] * reps)[:B16])
    ss = np.ascontiguousarray(np.concatenate([sl] * reps)[:B16])
    steps = 6 if dtype != "fp16" else 12
    with ca.Engine(spec, w, max_batch=B16, segment_len=L, n_slots=2, dtype=dtype, calibrate=calibrate) as e:
        for i in range(2):
            e.submit(i, xx, ss, beam_width=0, want_prob=True)
        for i in range(2):
            e.collect(i)
        t0 = time.perf_counter()
        for i in range(steps):
            if i >= 2:
                e.collect(i % 2)
            e.submit(i % 2, xx, ss, beam_width=0, want_prob=True)
        for i in range(2):
            e.collect(i)
        e.sync()
        return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=1100)
    ap.add_argument("--regimes", default="peaked,density,peaked-density")
    a = ap.parse_args()
    spec = ca.dna_default_spec()
    n = a.windows
    x, ln = pb.windows(JUMP * (n - 1) + 200, L, JUMP, 4711)
    sl = ca.seq_len_for_engine(ln, 1.0)
    out = {"windows": n, "regimes": {}, "ms_per_4096_windows": {}}
    for regime in a.regimes.split(","):
        if regime == "peaked":
            w = regimes.peaked_head(regimes.trained_like_weights(spec, x[:24], seed=5)[0])
        elif regime == "density":
            w = ca.synthetic_weights(spec, seed=1234)
            H = spec.hidden
            for k in w:
                if k.endswith("lstm_cell/bias"):
                    w[k][2 * H:3 * H] = -3.0
            with ca.Engine(spec, w, max_batch=n, segment_len=L) as e0:
                e0.infer(x, sl)
                h = e0.rnn_output()[:256]
            w = regimes.fit_emitting_head(w, h, x[:256], sl[:256], 43.875, hidden=H)
        else:
            w, _ = regimes.trained_like_weights(spec, x[:24], seed=5, forget_mean=-2.0)
            w = regimes.dense_head(spec, w, x[:32], sl[:32], 30)
        with ca.Engine(spec, w, max_batch=n, segment_len=L) as e32:
            ref = e32.infer(x, sl, want_logits=True)
        ref_rows = rows_of(ref, n)
        mask = np.arange(ref.logits.shape[1])[None, :] < sl[:, None]
        rec = {"bases_per_window_fp32": round(ref.decoded.values.shape[0] / float(n), 2), "modes": {}}
        for name, dtype, cal, lasth16 in MODES:
            if lasth16:
                os.environ["CHIRON_F16_LASTH16"] = "1"
            else:
                os.environ.pop("CHIRON_F16_LASTH16", None)
            with ca.Engine(spec, w, max_batch=n, segment_len=L, dtype=dtype, calibrate=cal) as e:
                r = e.infer(x, sl, want_logits=True)
            d = np.abs(r.logits - ref.logits)[mask]
            same = float(np.mean([np.array_equal(p, q) for p, q in zip(rows_of(r, n), ref_rows)]))
            rec["modes"][name] = {"identical_windows_frac": round(same, 4), "logits_mean_abs": float("%.3e" % d.mean()), "logits_p999_abs": float("%.3e" % np.quantile(d, 0.999))}
            print("%-15s %-36s identical %.4f  logits mean %.3g p99.9 %.3g" % (regime, name, same, d.mean(), np.quantile(d, 0.999)), flush=True)
            if regime == a.regimes.split(",")[0]:
                out["ms_per_4096_windows"][name] = round(time_4096(spec, w, x, sl, dtype, cal), 3)
                print("%-15s %-36s %.3f ms per 4096 windows" % ("", name, out["ms_per_4096_windows"][name]), flush=True)
        out["regimes"][regime] = rec
    os.environ.pop("CHIRON_F16_LASTH16", None)
    with ca.Engine(spec, ca.synthetic_weights(spec, seed=1234), max_batch=4096, segment_len=L, n_slots=2) as e:
        pass
    out["ms_per_4096_windows"]["fp32"] = round(time_4096(spec, ca.synthetic_weights(spec, seed=1234), x, sl, "fp32", False), 3)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "f16_frontier.json"), "w"), indent=1)
    print(json.dumps(out["ms_per_4096_windows"]))


if __name__ == "__main__":
    main()
