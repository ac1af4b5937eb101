#!/usr/bin/env python3
"""End-to-end `chiron call` throughput on synthetic reads (host pipeline + engine): N reads x 100k samples written
as .signal files, then chiron_amd.eval.evaluation with the DNA preset.
usage: e2e_bench.py [n_reads] [beam] [profile|timeline|-] [dtype] [batch]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chiron_amd as ca
from chiron_amd import eval as ce


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    beam = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    prof = len(sys.argv) > 3 and sys.argv[3] == "profile"
    timeline = len(sys.argv) > 3 and sys.argv[3] == "timeline"
    dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 1100
    d = tempfile.mkdtemp(prefix="e2e_")
    inp = os.path.join(d, "in")
    os.makedirs(inp)
    sig = ca.synthetic_signal(n_reads, 100000, seed=77)
    t0 = time.time()
    for i in range(n_reads):
        with open(os.path.join(inp, "read%04d.signal" % i), "w") as f:
            f.write(" ".join(str(int(v)) for v in sig[i]))
    t_write = time.time() - t0

    class F(object):
        input, output, model = inp, os.path.join(d, "out"), "synthetic"
        start, batch_size, segment_len, jump = 0, batch, 400, 390
        extension, concise, mode, recursive = "fastq", False, "dna", True
    F.beam = beam
    F.finish_procs = int(os.environ.get("E2E_FINISH_PROCS", "0"))
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=1234)
    with ca.Engine(spec, w, max_batch=batch, segment_len=400, n_slots=int(os.environ.get("E2E_SLOTS", "3")), max_beam=beam, dtype=dtype) as eng:
        ce.evaluation(F, engine=eng)          # warm-up (page cache, first launches)
        shutil.rmtree(F.output)
        pr = cProfile.Profile() if prof else None
        ev = []
        if timeline:   # where the wall time goes: engine calls on the main thread, reader waits, finisher spans
            import threading
            for obj, name in ((eng, "submit"), (eng, "collect"), (ce, "finish_read"), (ce.signal_io, "read_data_for_eval"),
                              (ce.assembly, "simple_assembly_qs"), (ce, "qs"), (ce, "write_output"), (ce, "index2base")):
                fn = getattr(obj, name)

                def wrapped(*a, _fn=fn, _name=name, **k):
                    t = time.time()
                    try:
                        return _fn(*a, **k)
                    finally:
                        ev.append((_name, threading.get_ident(), t, time.time()))
                setattr(obj, name, wrapped)
        t0 = time.time()
        if pr:
            pr.enable()
        out = ce.evaluation(F, engine=eng)
        if pr:
            pr.disable()
        dt = time.time() - t0
    windows = n_reads * 257
    print("reads %d  windows %d  beam %d  %s batch %d : %.2f s  -> %.0f windows/s, %.1f kbases/s (signal-normalised); input written in %.1f s"
          % (n_reads, windows, beam, dtype, batch, dt, windows / dt, windows * 390 / (4000 / 450.0) / 1000 / dt, t_write))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    if timeline:
        for name in ("submit", "collect", "finish_read", "read_data_for_eval", "simple_assembly_qs", "qs", "write_output", "index2base"):
            spans = [(a - t0, b - t0) for n, _, a, b in ev if n == name]
            tot = sum(b - a for a, b in spans)
            print("%-20s calls %4d  total %.3f s  mean %.2f ms  first start %.3f  last end %.3f  threads %d"
                  % (name, len(spans), tot, 1e3 * tot / max(1, len(spans)), min(a for a, _ in spans), max(b for _, b in spans),
                     len(set(t for n, t, _, _ in ev if n == name))))
        sub = sorted(a - t0 for n, _, a, b in ev if n == "submit")
        gaps = np.diff(sub)
        print("submit-to-submit gap: mean %.2f ms, p50 %.2f, p90 %.2f, max %.2f" % (1e3 * gaps.mean(), 1e3 * np.median(gaps),
                                                                                  1e3 * np.percentile(gaps, 90), 1e3 * gaps.max()))
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
