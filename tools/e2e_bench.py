#!/usr/bin/env python3
"""End-to-end `chiron call` throughput on synthetic reads (host pipeline + engine): N reads x 100k samples written
as .signal files, then chiron_amd.eval.evaluation with the DNA preset.
usage: e2e_bench.py [n_reads] [beam] [profile|timeline|-] [dtype] [batch] [signal|fast5|fast5-via-signal]
  signal            the reads are .signal text files (what extraction leaves under raw/)
  fast5             the reads are fast5 files (chunked + deflate int16, tests/h5_writer.py): the DIRECT path of `chiron call`
                    (native reader -> windows; raw/<name>.signal written for the output tree, never parsed back)
  fast5-via-signal  the same files through the reference's two passes: extract everything to raw/*.signal, then parse them"""
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


class F(object):      # the FLAGS of one `chiron call` (module level: extract()'s worker pool pickles it)
    start, segment_len, jump = 0, 400, 390
    extension, concise, mode, recursive = "fastq", False, "dna", True
    unit, idname, delimiter, test_number = False, False, "\n", None


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    beam = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    prof = len(sys.argv) > 3 and sys.argv[3] == "profile"
    timeline = len(sys.argv) > 3 and sys.argv[3] == "timeline"
    dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 1100
    kind = sys.argv[6] if len(sys.argv) > 6 else "signal"
    d = tempfile.mkdtemp(prefix="e2e_")
    inp = os.path.join(d, "in")
    os.makedirs(inp)
    sig = ca.synthetic_signal(n_reads, 100000, seed=77)
    t0 = time.time()
    if kind == "signal":
        for i in range(n_reads):
            with open(os.path.join(inp, "read%04d.signal" % i), "w") as f:
                f.write(" ".join(str(int(v)) for v in sig[i]))
    else:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import h5_writer
        for i in range(n_reads):
            h5_writer.write_multi_read_fast5(os.path.join(inp, "read%04d.fast5" % i), [("", "id-%d" % i, sig[i].astype(np.int16), None)],
                                             chunk=20000)
    t_write = time.time() - t0

    F.input, F.output, F.model = inp, os.path.join(d, "out"), "synthetic"
    F.batch_size = batch
    F.input_dir, F.output_dir = inp, os.path.join(d, "out")
    F.threads = int(os.environ.get("E2E_THREADS", "0"))
    F.beam = beam
    from chiron_amd import extract as ex

    def evaluate(eng):
        """one `chiron call` worth of host work + engine for the chosen input kind"""
        if kind == "signal":
            return ce.evaluation(F, engine=eng)
        if kind == "fast5":
            ex.prepare_folders(F)
            return ce.evaluation(F, engine=eng, fast5_files=ex.list_fast5(inp))
        ex.extract(F)                                       # the reference's two passes
        F.input = os.path.join(F.output, "raw")
        try:
            return ce.evaluation(F, engine=eng)
        finally:
            F.input = inp

    F.finish_procs = int(os.environ.get("E2E_FINISH_PROCS", "0"))
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=1234)
    with ca.Engine(spec, w, max_batch=batch, segment_len=400, n_slots=int(os.environ.get("E2E_SLOTS", "3")), max_beam=beam, dtype=dtype) as eng:
        evaluate(eng)                         # warm-up (page cache, first launches)
        shutil.rmtree(F.output)
        pr = cProfile.Profile() if prof else None
        ev = []
        if timeline:   # where the wall time goes: engine calls on the main thread, reader waits, finisher spans
            import threading
            for obj, name in ((eng, "submit"), (eng, "collect"), (ce, "finish_read_flat"), (ce, "finish_read"), (ce.signal_io, "read_data_for_eval"),
                              (ex, "extract_records"), (ce.signal_io, "window_signal"), (ce.assembly, "simple_assembly_qs"), (ce, "qs"),
                              (ce, "write_output"), (ce, "index2base")):
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
        out = evaluate(eng)
        if pr:
            pr.disable()
        dt = time.time() - t0
    windows = n_reads * 257
    print("input %s  reads %d  windows %d  beam %d  %s batch %d : %.2f s  -> %.0f windows/s, %.1f kbases/s (signal-normalised); input written in %.1f s"
          % (kind, n_reads, windows, beam, dtype, batch, dt, windows / dt, windows * 390 / (4000 / 450.0) / 1000 / dt, t_write))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    if timeline:
        for name in ("submit", "collect", "finish_read_flat", "finish_read", "read_data_for_eval", "extract_records", "window_signal",
                     "simple_assembly_qs", "qs", "write_output", "index2base"):
            spans = [(a - t0, b - t0) for n, _, a, b in ev if n == name]
            if not spans:
                continue
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
