#!/usr/bin/env python3
"""Secondary BASELINE.json configs (parity-test cases, not the headline bench line): timing only.
  configs[2]: RNA_default --mode rna, seg_len=500 jump=490 batch=400, CTC beam_width=50
  DNA beam:   DNA_default batch=1100 with the reference's preset beam widths (30 / 50)
"""
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca
from chiron_amd import signal_io


def run(name, spec, L, jump, B, beam, steps=20, dtype="fp32"):
    steps = int(os.environ.get("BENCH_STEPS", steps))     # longer timed regions for same-box A/B runs
    w = ca.synthetic_weights(spec, seed=1234)
    sig = ca.synthetic_signal(1, jump * (B - 1) + L, seed=5)[0]
    x, ln = signal_io.window_signal(sig, 0, jump, L)
    x, ln = x[:B], ln[:B]
    NS = int(os.environ.get("BENCH_SLOTS", "3"))      # what `chiron call` and bench.py keep in flight
    with ca.Engine(spec, w, max_batch=B, segment_len=L, n_slots=NS, max_beam=beam, dtype=dtype) as eng:
        sl = ca.seq_len_for_engine(ln, eng.ratio)
        for _ in range(2):
            eng.infer(x, sl, beam_width=beam)
        eng.sync()
        t0 = time.perf_counter()
        pend = [False] * NS
        nb = 0
        for i in range(steps):
            s = i % NS
            res = eng.collect(s) if pend[s] else None
            eng.submit(s, x, sl, beam_width=beam, want_prob=True)     # the slot gets its next batch before the host looks at the last
            pend[s] = True
            if res is not None:
                nb += res.decoded.values.shape[0]
        for s in range(NS):
            if pend[s]:
                nb += eng.collect(s).decoded.values.shape[0]
        dt = time.perf_counter() - t0
        eng.profile(True)
        eng.infer(x, sl, beam_width=beam)
        st = eng.profile_read()
        eng.profile(False)
    bases_per_window = jump / (4000.0 / 450.0) if "DNA" in name else jump / (3012.0 / 70.0)
    print(json.dumps({"config": name, "dtype": dtype, "batch": B, "segment_len": L, "jump": jump, "beam": beam, "T": spec.output_len(L),
                      "ms_per_batch": round(dt / steps * 1e3, 3), "windows_per_s": round(steps * B / dt, 1),
                      "decoded_bases_per_s": round(nb / dt, 1),
                      "kernels_ms": {k: round(v["total_ms"] / v["launches"], 4) for k, v in st.items()}}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "f16":  # BASELINE configs[4]: fp16 conv + LSTM, batch 4096 (and 1100 for comparison)
        run("DNA_default seg400 jump390 b4096 greedy f16", ca.dna_default_spec(), 400, 390, 4096, 0, dtype="fp16")
        run("DNA_default seg400 jump390 b1100 greedy f16", ca.dna_default_spec(), 400, 390, 1100, 0, dtype="fp16")
        run("DNA_default seg400 jump390 b4096 greedy f32", ca.dna_default_spec(), 400, 390, 4096, 0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "w2":  # f16 activations against exact (hi + lo) weights, at configs[4]'s batch and at the headline's
        run("DNA_default seg400 jump390 b4096 greedy fp16-w2", ca.dna_default_spec(), 400, 390, 4096, 0, dtype="fp16-w2")
        run("DNA_default seg400 jump390 b1100 greedy fp16-w2", ca.dna_default_spec(), 400, 390, 1100, 0, dtype="fp16-w2")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "split":  # fp32 values as hi/lo half pairs on the f16 matrix cores
        run("DNA_default seg400 jump390 b1100 greedy fp32-split", ca.dna_default_spec(), 400, 390, 1100, 0, dtype="fp32-split")
        run("DNA_default seg400 jump390 b1100 greedy fp32", ca.dna_default_spec(), 400, 390, 1100, 0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "batches":  # kernel time vs batch size (workgroup-count effects)
        for B in [int(v) for v in sys.argv[2:]]:
            run("DNA_default seg400 jump390 b%d greedy" % B, ca.dna_default_spec(), 400, 390, B, 0, steps=4)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "slots":  # batches in flight x decoder: `slots 3 4 5` (same-box comparison)
        for ns in sys.argv[2:]:
            os.environ["BENCH_SLOTS"] = ns
            for beam in (30, 0):
                run("DNA_default seg400 jump390 b1100 beam%d slots%s" % (beam, ns), ca.dna_default_spec(), 400, 390, 1100, beam)
        sys.exit(0)
    run("RNA_default seg500 jump490 b400 beam50", ca.rna_default_spec(), 500, 490, 400, 50, steps=100)   # 2 ms steps: 100 of them
    run("DNA_default seg400 jump390 b1100 beam30", ca.dna_default_spec(), 400, 390, 1100, 30)
    run("DNA_default seg400 jump390 b1100 beam50", ca.dna_default_spec(), 400, 390, 1100, 50)
    run("DNA_default seg400 jump390 b1100 greedy", ca.dna_default_spec(), 400, 390, 1100, 0)
