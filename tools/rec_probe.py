#!/usr/bin/env python3
"""Recurrence launch time (HIP events, single stream) against the batch size: how the resident rounds show.
    python tools/rec_probe.py 1024 1100 512 128          (env switches of the engine apply: CHIRON_LSTM_FIXED_ROLES, CHIRON_LSTM_PAIR)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca
from chiron_amd import signal_io


def probe(B, L=400, jump=390, reps=5):
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=1234)
    sig = ca.synthetic_signal(1, jump * (B - 1) + L, seed=5)[0]
    x, ln = signal_io.window_signal(sig, 0, jump, L)
    with ca.Engine(spec, w, max_batch=B, segment_len=L, n_slots=1, dtype=os.environ.get("PROBE_DTYPE", "fp32")) as eng:
        sl = ca.seq_len_for_engine(ln[:B], eng.ratio)
        for _ in range(2):
            eng.infer(x[:B], sl)
        eng.profile(True)
        for _ in range(reps):
            eng.infer(x[:B], sl)
        st = eng.profile_read()
        eng.profile(False)
    return {k: round(v["total_ms"] / v["launches"], 4) for k, v in st.items() if k in ("lstm_recurrence", "lstm_proj_dma", "lstm_proj0_dma", "conv_wino", "conv2a", "conv_dma", "conv_res")}


if __name__ == "__main__":
    for b in [int(a) for a in sys.argv[1:]] or [1024, 1100]:
        print(json.dumps({"batch": b, "env": {k: v for k, v in os.environ.items() if k.startswith("CHIRON_")}, **probe(b)}))
