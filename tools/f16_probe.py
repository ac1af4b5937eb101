#!/usr/bin/env python3
"""The fp16 engine at BASELINE configs[4] (DNA_default, batch 4096), six single-stream batches: the workload
tools/f16_profile.sh traces.  usage: f16_probe.py [batch] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca
from chiron_amd import signal_io

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
L, jump = 400, 390
spec = ca.dna_default_spec()
w = ca.synthetic_weights(spec, seed=1234)
sig = ca.synthetic_signal(1, jump * (B - 1) + L, seed=5)[0]
x, ln = signal_io.window_signal(sig, 0, jump, L)
with ca.Engine(spec, w, max_batch=B, segment_len=L, n_slots=1, dtype=dtype) as eng:
    sl = ca.seq_len_for_engine(ln[:B], eng.ratio)
    for _ in range(6):
        eng.infer(x[:B], sl)
