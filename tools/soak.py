#!/usr/bin/env python3
"""Soak run: the same ragged batches through an engine again and again, on all slots, interleaved with other batch sizes,
checking that every repetition reproduces the first result bit for bit (a data race in a kernel shows up as a rare
mismatch) and that device memory does not grow.   usage: soak.py [seconds] [dtype] [max_batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
    import torch
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=1234)
    L = 400
    rng = np.random.RandomState(5)
    cases = []
    for B in sorted(set([max_batch, max(1, max_batch // 2 + 3), 17, 1])):
        x = ca.synthetic_signal(1, B * L, seed=B)[0].reshape(B, L).copy()
        ln = rng.randint(0, L + 1, size=B)
        ln[rng.randint(0, B)] = L
        ln[: max(1, B // 3)] = L                    # mostly full windows, a ragged tail
        for b in range(B):
            x[b, ln[b]:] = 0
        cases.append((x, ln))
    slots = 3
    free0 = None
    with ca.Engine(spec, w, max_batch=max_batch, segment_len=L, n_slots=slots, max_beam=30, dtype=dtype) as eng:
        ref = []
        for x, ln in cases:
            sl = ca.seq_len_for_engine(ln, eng.ratio)
            r = eng.infer(x, sl, want_prob=True, want_logits=True)
            rb = eng.infer(x, sl, beam_width=30)
            ref.append((sl, r.logits.copy(), r.decoded.values.copy(), r.prob_logits.copy(), rb.decoded.values.copy(), rb.log_prob.copy()))
        t0 = time.time()
        n = bad = 0
        while time.time() - t0 < seconds:
            if free0 is None and n >= 4 * len(cases):   # after the first rounds: the runtime's own pools have been created
                torch.cuda.synchronize()
                free0 = torch.cuda.mem_get_info()[0]
            order = rng.permutation(len(cases))
            pend = []
            for k, ci in enumerate(order):           # all slots busy with different batch sizes at once
                x, _ = cases[ci]
                beam = 30 if (n + k) % 4 == 3 else 0
                eng.submit(k % slots, x, ref[ci][0], beam_width=beam, want_prob=True, want_logits=beam == 0)
                pend.append((k % slots, ci, beam))
                if len(pend) == slots or k + 1 == len(order):
                    for s, cj, bm in pend:
                        r = eng.collect(s)
                        if bm == 0:
                            ok = (np.array_equal(r.logits.view(np.uint32), ref[cj][1].view(np.uint32)) and np.array_equal(r.decoded.values, ref[cj][2])
                                  and np.array_equal(r.prob_logits, ref[cj][3]))
                        else:
                            ok = np.array_equal(r.decoded.values, ref[cj][4]) and np.array_equal(r.log_prob, ref[cj][5])
                        bad += 0 if ok else 1
                        n += 1
                    pend = []
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
    print("soak %s max_batch %d: %d batches in %.0f s, %d mismatches, device memory change %d bytes" % (dtype, max_batch, n, time.time() - t0, bad, free0 - free1))
    return 1 if bad or free0 != free1 else 0


if __name__ == "__main__":
    sys.exit(main())
