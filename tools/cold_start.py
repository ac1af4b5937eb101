#!/usr/bin/env python3
"""Cold-start budget of one rank of `chiron call`: from process start to the first collected batch, stage by stage.

BASELINE configs[3] gives every GPU 1250 reads = 321 k windows: 3.1 s of fp32 compute, 0.6 s behind the fp16 engine -- so what a
rank spends BEFORE its first batch decides the end-to-end 8-GPU figure.  A child process (fresh interpreter, nothing cached in
this process) times:

  interpreter          exec of python3 until the first line of the script (measured by the parent's clock)
  import numpy
  import chiron_amd    (pulls in no torch: only bench.py / torch.distributed.run launches import it)
  load library         dlopen of libchiron_amd.so (+ the HIP runtime libraries it links)
  model                load_model: model.json + checkpoint index (or seeded synthetic weights when the data blob is absent)
  engine create        chiron_engine_create, with its own breakdown from CHIRON_TRACE_CREATE=1 (HIP runtime + context + code objects,
                       weight preparation + upload, slot buffers)
  first batch          submit + collect of one full batch (first launch of every kernel)
  second batch         the steady state, for comparison

    python tools/cold_start.py [--batch 1100] [--dtype fp32] [--slots 3] [--repeat 3]  ->  gpurun_out/cold_start.json
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import time
t_first_line = time.time()
import json, os, sys
sys.path.insert(0, %(root)r)
stages = {}
def stage(name, t0):
    stages[name] = round((time.time() - t0) * 1e3, 1)
t = time.time(); import numpy as np; stage("import_numpy_ms", t)
t = time.time(); import chiron_amd as ca; stage("import_chiron_amd_ms", t)
from chiron_amd import _lib
t = time.time(); _lib.load(); stage("load_library_ms", t)
model = os.path.join(%(root)r, "chiron_amd", "model", "DNA_default")
t = time.time(); spec, w, _ = ca.load_model(model, allow_synthetic=True); stage("load_model_ms", t)
t = time.time(); blob = spec.pack(w); stage("pack_weights_ms", t)
B, L = %(batch)d, 400
x = (np.random.RandomState(0).randn(B, L) * 60 + 500).astype(np.float32)
sl = np.full(B, L, dtype=np.int32)
t = time.time(); eng = ca.Engine(spec, blob, max_batch=B, segment_len=L, n_slots=%(slots)d, dtype=%(dtype)r); stage("engine_create_ms", t)
t = time.time(); eng.infer(x, sl); stage("first_batch_ms", t)
t = time.time(); eng.infer(x, sl); stage("second_batch_ms", t)
stages["torch_imported"] = "torch" in sys.modules
stages["t_first_line"] = t_first_line
stages["t_ready_for_second_batch"] = t
print("STAGES " + json.dumps(stages))
'''


def once(batch, dtype, slots):
    code = CHILD % {"root": ROOT, "batch": batch, "slots": slots, "dtype": dtype}
    env = dict(os.environ, CHIRON_TRACE_CREATE="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    st = json.loads([l for l in r.stdout.split("\n") if l.startswith("STAGES ")][0][7:])
    st["interpreter_start_ms"] = round((st.pop("t_first_line") - t0) * 1e3, 1)
    st["process_start_to_first_batch_collected_ms"] = round((st.pop("t_ready_for_second_batch") - t0) * 1e3, 1)
    for l in r.stderr.split("\n"):
        if l.startswith("chiron_engine_create: "):
            st["engine_create_breakdown"] = json.loads(l[len("chiron_engine_create: "):])
    return st


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1100)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    runs = [once(a.batch, a.dtype, a.slots) for _ in range(a.repeat)]
    out = {"batch": a.batch, "dtype": a.dtype, "slots": a.slots, "runs": runs,
           "note": "run 1 is the coldest (page cache of the library and of numpy on a fresh box); the last run is what every later rank start costs"}
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "cold_start_%s_%d.json" % (a.dtype, a.batch)), "w"), indent=1)
    for i, r in enumerate(runs):
        print("run %d: process start -> first batch collected %.0f ms | interpreter %.0f, numpy %.0f, chiron_amd %.0f, library %.0f, model %.0f, pack %.0f, "
              "create %.0f %s, first batch %.0f, second batch %.1f" % (
                  i + 1, r["process_start_to_first_batch_collected_ms"], r["interpreter_start_ms"], r["import_numpy_ms"], r["import_chiron_amd_ms"],
                  r["load_library_ms"], r["load_model_ms"], r["pack_weights_ms"], r["engine_create_ms"], json.dumps(r.get("engine_create_breakdown", {})),
                  r["first_batch_ms"], r["second_batch_ms"]))


if __name__ == "__main__":
    main()
