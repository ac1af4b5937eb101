#!/usr/bin/env python
"""getcnnfeature (cnn.py:334-371) of the fp32 engine against the float64 oracle on trained-checkpoint-like weights, for every
convolution form the engine has (A/B switches of INTEGRATION.md): which form costs how much accuracy.  tools/parity_budget.py
showed that on these weights the logits' deviation is the CNN features' rounding error amplified by the recurrent stack.

  python tools/cnn_forms_error.py [weight seeds, default 5,6,7,8]   ->  gpurun_out/cnn_forms_error.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import chiron_amd as ca            # noqa: E402
from oracle import nn_oracle       # noqa: E402
import regimes                     # noqa: E402
import parity_budget as pb         # noqa: E402

FORMS = (("default", {}), ("no-table", {"CHIRON_NO_PWL": "1"}), ("no-winograd", {"CHIRON_NO_WINOGRAD": "1"}),
         ("winograd-f2", {"CHIRON_WINOGRAD_F2": "1"}), ("no-stream32", {"CHIRON_NO_STREAM32": "1"}),
         ("all-tiled-gemm", {"CHIRON_NO_PWL": "1", "CHIRON_NO_WINOGRAD": "1", "CHIRON_NO_STREAM32": "1"}))
SWITCHES = ("CHIRON_NO_PWL", "CHIRON_NO_WINOGRAD", "CHIRON_WINOGRAD_F2", "CHIRON_NO_STREAM32")


def main():
    seeds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5,6,7,8").split(",")]
    out = []
    for topo in ("dna", "rna"):
        spec = ca.dna_default_spec() if topo == "dna" else ca.rna_default_spec()
        L, jump = (400, 390) if topo == "dna" else (500, 490)
        for k, ws in enumerate(seeds):
            x, ln = pb.windows(jump * 23 + 200, L, jump, 67 + 10 * k)
            w, _ = regimes.trained_like_weights(spec, x[:24], seed=ws)
            sd = spec.to_dict()
            f64 = nn_oracle.cnn_forward(x.astype(np.float64), sd, w)
            f32 = nn_oracle.cnn_forward(x.astype(np.float32), sd, {kk: v.astype(np.float32) for kk, v in w.items()})
            row = {"topology": topo, "weight_seed": ws, "numpy_fp32": pb.stats(f32, f64)}
            for name, env in FORMS:
                for s in SWITCHES:
                    os.environ.pop(s, None)
                os.environ.update(env)
                with ca.Engine(spec, w, max_batch=x.shape[0], segment_len=L) as eng:
                    eng.infer(x, ca.seq_len_for_engine(ln, eng.ratio))
                    row[name] = pb.stats(eng.features(), f64)
            for s in SWITCHES:
                os.environ.pop(s, None)
            out.append(row)
            print(topo, ws, " ".join("%s %.3g/%.3g" % (n, row[n]["max"], row[n]["rms"]) for n in ["numpy_fp32"] + [f[0] for f in FORMS]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cnn_forms_error.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
