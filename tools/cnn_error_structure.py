#!/usr/bin/env python
"""Structure of getcnnfeature's rounding error (cnn.py:334-371) on trained-checkpoint-like weights: where it is born, what it looks
like, and what the recurrent stack makes of it.  Round 4's budget (tools/parity_budget.py) showed that the logits' deviation is
the features' error amplified by the stack, and one RNA weight set (seed 6) amplified the engine's error 20 x against 3.6 x for the
float32 numpy restatement's at the same rms.  Is the engine's error STRUCTURED (a per-channel constant, correlated along the
sequence), or is the statistic noisy?  Per topology and weight set this prints and writes:

  * per residual block (engines built with 1, 2, 3 blocks: the descriptor is data-driven): total error against the float64 oracle
    and the error BORN in the block (the float64 block applied to the implementation's own previous block output), for the engine
    in every convolution form (table / Winograd / streaming / tiled GEMM switches of INTEGRATION.md) and for float32 restatements
    in three summation orders: `natural` (oracle/nn_oracle.py: BLAS, one K = C chain per tap, BN applied to the rounded sum),
    `folded` (BN folded into the filters as the engine does, BLAS chains), `chain` (folded, ONE sequential fmaf chain over the
    whole K starting from the shift: what an MFMA accumulator does, /opt/skills/guides: "exact f32 == an fmaf chain");
  * for the features' error e[b, t, c]: its rms, the rms of its per-channel mean over all frames (DC), the lag-1 autocorrelation of
    the rest along t, and what each part costs AT THE LOGITS with the float64 oracle doing everything behind the features;
  * the same cost for WHITE noise of the same rms, five draws: the spread a structure-free error of that size produces;
  * per window: the largest logit deviation -- one ill-conditioned window carries a weight set's whole max-norm.

  python tools/cnn_error_structure.py [--seeds 5,6,7,8] [--topologies dna,rna] [--no-engine]  ->  gpurun_out/cnn_error_structure.json
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

F32 = np.float32
FORMS = (("default", {}), ("no-table", {"CHIRON_NO_PWL": "1"}), ("no-winograd", {"CHIRON_NO_WINOGRAD": "1"}),
         ("no-stream32", {"CHIRON_NO_STREAM32": "1"}),
         ("all-tiled-gemm", {"CHIRON_NO_PWL": "1", "CHIRON_NO_WINOGRAD": "1", "CHIRON_NO_STREAM32": "1"}))
SWITCHES = ("CHIRON_NO_PWL", "CHIRON_NO_WINOGRAD", "CHIRON_WINOGRAD_F2", "CHIRON_NO_STREAM32")


def rms(a):
    return float(np.sqrt((np.asarray(a, dtype=np.float64) ** 2).mean()))


fold, fma32, conv_chain, block_f32, restatement_blocks = pb.fold, pb.fma32, pb.conv_chain, pb.block_f32, pb.restatement_blocks


def engine_blocks(spec, w, x, ln, L):
    out = []
    for n in range(1, len(spec.blocks) + 1):
        sp = ca.ModelSpec(spec.blocks[:n], spec.rnn_kind, spec.rnn_layers, spec.hidden, spec.classes, spec.bn_mode, spec.stem)
        keep = set(sp.blob_layout())
        with ca.Engine(sp, {k: v for k, v in w.items() if k in keep}, max_batch=x.shape[0], segment_len=L) as eng:
            eng.infer(x, ca.seq_len_for_engine(ln, eng.ratio))
            out.append(eng.features())
    return out


def structure(e):
    dc = e.mean(axis=(0, 1), keepdims=True)
    ac = e - dc
    return dc, ac, {"rms": rms(e), "max": float(np.abs(e).max()), "dc_rms": rms(dc), "ac_rms": rms(ac),
                    "lag1": float((ac[:, 1:] * ac[:, :-1]).mean() / max((ac * ac).mean(), 1e-300))}


def study(topology, k, ws, with_engine, n_windows=24):
    spec = ca.dna_default_spec() if topology == "dna" else ca.rna_default_spec()
    L, jump = (400, 390) if topology == "dna" else (500, 490)
    x, ln = pb.windows(jump * (n_windows - 1) + 200, L, jump, 67 + 10 * k)
    ln = ln.copy()
    ln[2], ln[5] = L // 3, 0
    w, _ = regimes.trained_like_weights(spec, x[:24], seed=ws)
    sd = spec.to_dict()
    w64 = {kk: np.asarray(v, dtype=np.float64) for kk, v in w.items()}
    blocks64, p = [], x.astype(np.float64)[:, :, None]
    for blk in sd["cnn"]:
        p = nn_oracle.residual_layer(p, w64, blk, "population")
        blocks64.append(p)
    T = blocks64[-1].shape[1]
    sl = ca.seq_len_for_engine(ln, L / float(T))
    logits64 = pb.propagate_to_logits(spec, w, "features", blocks64[-1], sl)
    fmask = (np.arange(T)[None, :] < np.asarray(sl)[:, None])

    def at_logits(fea):
        d = np.abs(pb.propagate_to_logits(spec, w, "features", fea, sl) - logits64).max(axis=-1) * fmask
        return {"max": float(d.max()), "rms": float(np.sqrt((d[fmask] ** 2).mean())), "worst_window": int(d.max(axis=1).argmax()),
                "second_window_max": float(np.sort(d.max(axis=1))[-2])}

    impls = {"numpy_fp32_" + o: restatement_blocks(x, sd, w, o) for o in ("natural", "folded", "chain")}
    if with_engine:
        for name, env in FORMS:
            for s in SWITCHES:
                os.environ.pop(s, None)
            os.environ.update(env)
            impls["engine_" + name] = engine_blocks(spec, w, x, ln, L)
        for s in SWITCHES:
            os.environ.pop(s, None)
    rep = {"topology": topology, "weight_seed": ws, "signal_seed": 67 + 10 * k, "windows": n_windows, "feature_scale_rms": rms(blocks64[-1]),
           "implementations": {}}
    for name, bl in impls.items():
        r = {"blocks": []}
        prev = x.astype(np.float64)[:, :, None]
        for i, blk in enumerate(sd["cnn"]):
            local_ref = nn_oracle.residual_layer(prev, w64, blk, "population")
            r["blocks"].append({"total_rms": rms(bl[i] - blocks64[i]), "total_max": float(np.abs(bl[i] - blocks64[i]).max()),
                                "local_rms": rms(bl[i] - local_ref), "local_max": float(np.abs(bl[i] - local_ref).max())})
            prev = bl[i].astype(np.float64)
        e = bl[-1].astype(np.float64) - blocks64[-1]
        dc, ac, st = structure(e)
        r["features"] = st
        r["at_logits"] = {"full": at_logits(blocks64[-1] + e), "dc_only": at_logits(blocks64[-1] + dc), "ac_only": at_logits(blocks64[-1] + ac)}
        rep["implementations"][name] = r
        print("%s %d %-22s born/block %s | features rms %.3g (dc %.3g, lag1 %+.2f) | at logits full %.3g / %.3g  dc %.3g  ac %.3g  (window %d; next %.3g)" % (
            topology, ws, name, " ".join("%.3g" % b["local_rms"] for b in r["blocks"]), st["rms"], st["dc_rms"], st["lag1"],
            r["at_logits"]["full"]["max"], r["at_logits"]["full"]["rms"], r["at_logits"]["dc_only"]["max"], r["at_logits"]["ac_only"]["max"],
            r["at_logits"]["full"]["worst_window"], r["at_logits"]["full"]["second_window_max"]), flush=True)
    # structure-free noise of the natural restatement's size: the spread of the statistic itself
    sigma = rep["implementations"]["numpy_fp32_natural"]["features"]["rms"]
    rng = np.random.RandomState(12345 + ws)
    rep["white_noise_of_numpy_fp32_rms"] = [at_logits(blocks64[-1] + rng.normal(0.0, sigma, blocks64[-1].shape)) for _ in range(5)]
    print("%s %d white noise of rms %.3g at the logits, five draws (max / rms): %s" % (
        topology, ws, sigma, "  ".join("%.3g / %.3g" % (d["max"], d["rms"]) for d in rep["white_noise_of_numpy_fp32_rms"])), flush=True)
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="5,6,7,8")
    ap.add_argument("--topologies", default="dna,rna")
    ap.add_argument("--no-engine", action="store_true", help="float32 restatements only (runs without a GPU)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cnn_error_structure.json"))
    a = ap.parse_args()
    out = [study(t, ws - 5, ws, not a.no_engine) for t in a.topologies.split(",") for ws in [int(v) for v in a.seeds.split(",")]]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
