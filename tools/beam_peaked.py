#!/usr/bin/env python3
"""Beam-search cost on PEAKED posteriors.  The seeded synthetic weights give flat posteriors (beam 30 decodes 88 bases per
window where greedy decodes 6), the worst case for the event-driven beam kernel; a trained CTC model is blank-dominated
with short confident base spikes.  This feeds such logits -- a random base roughly every 9 frames (450 bases/s at 4 kHz
over the 400 frames of a window), blank 0.95 elsewhere, spikes of 1-2 frames at 0.6-0.97, some doubtful frames -- through the
decode-only entry (chiron_engine_decode, profiling bucket of the beam kernel) next to the engine's own logits.
    usage: beam_peaked.py [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca
from chiron_amd import signal_io


def peaked_logits(B, T, rng):
    p = np.full((B, T, 5), 0.0125, dtype=np.float64)
    p[:, :, 4] = 0.95
    for b in range(B):
        t = int(rng.randint(0, 9))
        while t < T:
            base = int(rng.randint(0, 4))
            conf = rng.uniform(0.6, 0.97)
            for d in range(int(rng.randint(1, 3))):
                if t + d < T:
                    row = np.full(5, (1.0 - conf) / 4)
                    row[base] = conf
                    p[b, t + d] = row
            if rng.rand() < 0.15 and t + 3 < T:         # a doubtful frame: two bases and blank compete
                alt = int(rng.randint(0, 4))
                row = np.full(5, 0.02)
                row[alt], row[base], row[4] = 0.3, 0.25, 0.41
                p[b, t + 3] = row
            t += int(rng.randint(5, 14))
    return np.log(p).astype(np.float32)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
    L, jump = 400, 390
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=1234)
    rng = np.random.RandomState(9)
    with ca.Engine(spec, w, max_batch=B, segment_len=L, n_slots=1, max_beam=50) as eng:
        T = eng.T
        sl = np.full(B, T, dtype=np.int32)
        lg = peaked_logits(B, T, rng)
        sig = ca.synthetic_signal(1, jump * (B - 1) + L, seed=5)[0]
        x, ln = signal_io.window_signal(sig, 0, jump, L)
        own = eng.infer(x[:B], ca.seq_len_for_engine(ln[:B], eng.ratio), want_logits=True).logits.copy()
        for name, logits in (("peaked (trained-model-like)", lg), ("engine logits of the synthetic weights", own)):
            for beam in (0, 30, 50):
                eng.decode(logits, sl, beam_width=beam)
                eng.profile(True)
                for _ in range(3):
                    r = eng.decode(logits, sl, beam_width=beam)
                st = eng.profile_read()
                eng.profile(False)
                ms = sum(v["total_ms"] for k, v in st.items() if "beam" in k or "greedy" in k or "sparse" in k) / 3.0
                print("%-42s beam %2d: decode kernels %.3f ms per %d-window batch, %.1f bases per window"
                      % (name, beam, ms, B, r.decoded.values.shape[0] / float(B)))


if __name__ == "__main__":
    main()
