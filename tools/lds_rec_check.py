#!/usr/bin/env python3
"""The recurrence with one W_hh tile per wave in the LDS (<= 128 registers: two workgroups per CU; build/libchiron_lstm_lds.so) against the
product's lstm32w2_kernel: bit-identical logits (DNA and RNA, ragged batch), then the launch time alone at batch 1100 and 2200."""
import os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:          # child: compute and dump
    import chiron_amd as ca
    from chiron_amd import signal_io
    out = {}
    for topo in ("dna", "rna"):
        spec = ca.dna_default_spec() if topo == "dna" else ca.rna_default_spec()
        L, jump = (400, 390) if topo == "dna" else (500, 490)
        w = ca.synthetic_weights(spec, seed=21)
        x, ln = signal_io.window_signal(ca.synthetic_signal(1, jump * 136 + 123, seed=3)[0], 0, jump, L)
        ln = ln.copy(); ln[3], ln[40] = L // 3, 0
        with ca.Engine(spec, w, max_batch=len(ln), segment_len=L) as e:
            out[topo] = e.infer(np.array(x), ca.seq_len_for_engine(ln, e.ratio), want_logits=True).logits
    np.savez(sys.argv[1], **out)
    sys.exit(0)
lib = os.path.join(ROOT, "build", "libchiron_lstm_lds.so")
for name, env in (("product", {}), ("lds", {"CHIRON_AMD_LIB": lib})):
    subprocess.check_call([sys.executable, __file__, "/tmp/lds_%s.npz" % name], env=dict(os.environ, **env))
a, b = np.load("/tmp/lds_product.npz"), np.load("/tmp/lds_lds.npz")
for k in a.files:
    print(k, "bit-identical:", bool(np.array_equal(a[k], b[k])), "max diff", float(np.abs(a[k] - b[k]).max()))
for name, env in (("product", {}), ("lds", {"CHIRON_AMD_LIB": lib}), ("product", {}), ("lds", {"CHIRON_AMD_LIB": lib})):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rec_probe.py"), "1100", "2200", "4096"], env=dict(os.environ, **env), capture_output=True, text=True)
    for l in r.stdout.splitlines():
        d = json.loads(l); print(name, d["batch"], d["lstm_recurrence"])
