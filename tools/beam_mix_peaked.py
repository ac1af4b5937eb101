#!/usr/bin/env python3
"""Beam search inside the three-batch pipeline on PEAKED posteriors (trained-checkpoint-like weights, tests/regimes.py, class
layer x 4, blank ahead by default): ms per 1100-window batch with beam 30 / 50 and greedy.  The synthetic weights of the bench
give flat posteriors (few events per frame); a trained model gives peaked ones (an insertion in nearly every frame)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import chiron_amd as ca
from chiron_amd import signal_io
import regimes


def main():
    steps = int(os.environ.get("BENCH_STEPS", "120"))
    spec = ca.dna_default_spec()
    L, jump, B = 400, 390, 1100
    sig = ca.synthetic_signal(1, jump * (B - 1) + L, seed=5)[0]
    x, ln = signal_io.window_signal(sig, 0, jump, L)
    x, ln = x[:B], ln[:B]
    w, _ = regimes.trained_like_weights(spec, x[:24], seed=5)
    w = dict(w)
    w["rnn_fnn_layer/weights_class"] = (w["rnn_fnn_layer/weights_class"] * 4.0).astype(np.float32)
    bc = w["rnn_fnn_layer/bias_class"].copy()
    bc[4] += 2.0
    w["rnn_fnn_layer/bias_class"] = bc
    NS = 3
    with ca.Engine(spec, w, max_batch=B, segment_len=L, n_slots=NS, max_beam=50) as eng:
        sl = ca.seq_len_for_engine(ln, eng.ratio)
        for beam in (30, 50, 0):
            for _ in range(2):
                eng.infer(x, sl, beam_width=beam)
            eng.sync()
            t0 = time.perf_counter()
            pend = [False] * NS
            nbase = 0
            for i in range(steps):
                s = i % NS
                res = eng.collect(s) if pend[s] else None
                eng.submit(s, x, sl, beam_width=beam, want_prob=True)
                pend[s] = True
                if res is not None:
                    nbase += res.decoded.values.shape[0]
            for s in range(NS):
                if pend[s]:
                    nbase += eng.collect(s).decoded.values.shape[0]
            dt = time.perf_counter() - t0
            print("peaked posteriors, three batches in flight, beam %2d: %.3f ms per batch, %.1f bases per window" % (beam, dt / steps * 1e3, nbase / steps / B))


if __name__ == "__main__":
    main()
