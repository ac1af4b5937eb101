"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/chiron_amd.h
declares; argument validation and the host-only entry point (chiron_assemble) work without a GPU.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import chiron_amd as ca
from chiron_amd import _lib, model as model_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "chiron_amd.h")).read()
    declared = set(re.findall(r"\b(chiron_[a-z0-9_]+)\s*\(", header))
    declared -= {"chiron_status"}
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, "binding and header disagree: %s" % (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.chiron_abi_version() == _lib.ABI_VERSION == 7
    assert lib.chiron_build_flags() == 0          # the product library is never a timing build


def test_timing_variants_cannot_ship_silently(built, tmp_path):
    """csrc/timing_variants.h: a kernel source compiled with a timing switch (parts of the kernel off, garbage results) but
    without CHIRON_TIMING_BUILD does not compile; with it, the library reports CHIRON_BUILD_TIMING and the binding refuses it
    unless CHIRON_ALLOW_TIMING_BUILD=1.  Every garbage-result switch of the sources is declared in that one header."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "chiron_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    # stream32.hip has no timing switch of its own: compile a two-line source that includes the header
    src = tmp_path / "probe.hip"
    src.write_text('#include "timing_variants.h"\nint probe_value = CHIRON_SENS;\n')
    bad = subprocess.run(base + ["-I", csrc, "-DCHIRON_SENS=1", "-c", str(src), "-o", str(tmp_path / "bad.o")], capture_output=True, text=True)
    assert bad.returncode != 0 and "computes garbage" in bad.stderr
    ok = subprocess.run(base + ["-I", csrc, "-DCHIRON_SENS=1", "-DCHIRON_TIMING_BUILD", "-c", str(src), "-o", str(tmp_path / "mark.o")],
                        capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    objs = [os.path.join(csrc, o) for o in sorted(os.listdir(csrc)) if o.endswith(".o")]
    lib_path = str(tmp_path / "libtiming.so")
    ln = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, str(tmp_path / "mark.o")] + objs + ["-lz", "-ldl"],
                        capture_output=True, text=True)
    assert ln.returncode == 0, ln.stderr
    code = "import sys; sys.path.insert(0, %r); from chiron_amd import _lib; print(_lib.load().chiron_build_flags())" % ROOT
    env = dict(os.environ, CHIRON_AMD_LIB=lib_path)
    env.pop("CHIRON_ALLOW_TIMING_BUILD", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "TIMING build" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, CHIRON_ALLOW_TIMING_BUILD="1"), capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "1", r.stderr
    # no garbage-result switch outside the header: every "#if CHIRON_<X>" of the kernel sources names a macro the header declares
    header = open(os.path.join(csrc, "timing_variants.h")).read()
    declared = set(re.findall(r"#ifndef (CHIRON_[A-Z0-9_]+)", header))
    assert declared == {"CHIRON_SENS", "CHIRON_W32_VARIANT", "CHIRON_F16F_VARIANT", "CHIRON_S16_VARIANT"}
    product_forms = {"CHIRON_GATE_MATH", "CHIRON_WINO_ROWMAJOR_STORES", "CHIRON_BEAM_PRIO"}   # compile-time choices that compute correct results (lstm.hip, wino.hip, beam.hip)
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".cpp", ".h")) and name != "timing_variants.h":
            text = open(os.path.join(csrc, name)).read()
            used = set(re.findall(r"#\s*(?:if|elif|ifdef|ifndef)[^\n]*?\b(CHIRON_[A-Z0-9_]+)", text))
            used -= {"CHIRON_AMD_H", "CHIRON_TIMING_BUILD"}
            assert used <= declared | product_forms, (name, used - declared)
            if used & declared:
                assert '#include "timing_variants.h"' in text, name


def test_weights_size_and_validation(built):
    lib = _lib.load()
    for spec, n in ((ca.dna_default_spec(), 1827333),):
        d = spec.to_c()
        out = C.c_size_t()
        assert lib.chiron_weights_size(C.byref(d), C.byref(out)) == _lib.OK
        assert out.value == n == spec.pack(ca.synthetic_weights(spec, 1)).size
    d = ca.rna_default_spec().to_c()
    out = C.c_size_t()
    assert lib.chiron_weights_size(C.byref(d), C.byref(out)) == _lib.OK
    assert out.value == ca.rna_default_spec().pack(ca.synthetic_weights(ca.rna_default_spec(), 1)).size
    bad = ca.dna_default_spec().to_c()
    bad.blocks[1].in_channels = 7
    assert lib.chiron_weights_size(C.byref(bad), C.byref(out)) == _lib.ERR_INVALID
    assert b"in_channels" in lib.chiron_last_error()
    bad = ca.dna_default_spec().to_c()
    bad.n_blocks = 0
    assert lib.chiron_weights_size(C.byref(bad), C.byref(out)) == _lib.ERR_INVALID


def test_engine_create_fails_loudly_without_gpu_or_with_bad_args(built):
    import torch
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, 1)
    with pytest.raises(ValueError):
        ca.Engine(spec, spec.pack(w)[:-1], max_batch=4, segment_len=400)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.ChironError) as ei:
            ca.Engine(spec, w, max_batch=4, segment_len=400)
        assert ei.value.status == _lib.ERR_DEVICE and "no CPU fallback" in str(ei.value)


def test_pack_rejects_wrong_shapes():
    spec = ca.dna_default_spec()
    w = dict(ca.synthetic_weights(spec, 1))
    w["rnn_fnn_layer/bias"] = np.zeros(99, np.float32)
    with pytest.raises(ValueError):
        spec.pack(w)
    del w["rnn_fnn_layer/bias"]
    with pytest.raises(KeyError):
        spec.pack(w)


def test_spec_from_variables_is_data_driven():
    for spec in (ca.dna_default_spec(), ca.rna_default_spec()):
        got = ca.spec_from_variables(dict(spec.variables()))
        assert got.blocks == spec.blocks and got.rnn_kind == spec.rnn_kind and got.hidden == 100
        assert got.rnn_layers == 3 and got.bn_mode == "population" and got.classes == 5


def test_chiron_assemble_errors(built):
    lib = _lib.load()
    bases = np.zeros(4, np.uint8)
    off = np.asarray([0, 2, 4], np.int64)
    n = C.c_int64()
    st = lib.chiron_assemble(bases.ctypes.data, off.ctypes.data, 2, None, 7, 0.2, 1.0, None, None, 0, C.byref(n))
    assert st == _lib.ERR_INVALID
    counts = np.zeros((4, 1))
    st = lib.chiron_assemble(bases.ctypes.data, off.ctypes.data, 2, None, _lib.KERNAL_STICK, 0.2, 1.0, counts.ctypes.data, None, 1, C.byref(n))
    assert st == _lib.ERR_OVERFLOW and n.value == 4


def test_stem_topology_contract(built):
    """HEAD RNA_model2 / RNA_model3 (cnn.py:454-476): stem conv + BN + ReLU in front of three 256-channel blocks.
    Variable manifest, blob size through the C ABI, topology recovery from variable shapes, frame count."""
    import ctypes as C
    lib = _lib.load()
    for model, (k, stride, T500) in {"rna_model2": (9, 5, 100), "rna_model3": (14, 7, 72)}.items():
        spec = ca.rna_head_spec(model)
        v = spec.variables()
        names = list(v)
        assert names[0] == "conv_layer/conv1/weights" and v[names[0]] == (1, k, 1, 256)
        assert names[1:5] == ["conv_layer/conv1_bn/" + n for n in ("scale", "offset", "pop_mean", "pop_var")]
        assert v["res_layer1/branch1/conv1/weights"] == (1, 1, 256, 256) and "res_layer1/branch1/conv1_bn/scale" in v
        assert spec.output_len(500) == T500
        w = ca.synthetic_weights(spec, seed=3)
        blob = spec.pack(w)
        n = C.c_size_t()
        d = spec.to_c()
        assert (d.stem_k, d.stem_stride, d.stem_channels) == (k, stride, 256)
        assert lib.chiron_weights_size(C.byref(d), C.byref(n)) == _lib.OK and n.value == blob.size
        back = model_mod.spec_from_variables({name: a.shape for name, a in w.items()})
        assert back.stem == spec.stem and back.blocks == spec.blocks and back.rnn_kind == "multi"
    d = ca.rna_head_spec().to_c()
    d.stem_channels = 128                                        # does not feed the first block
    assert lib.chiron_weights_size(C.byref(d), C.byref(n)) == _lib.ERR_INVALID
    with pytest.raises(ValueError):
        ca.ModelSpec(ca.dna_default_spec().blocks, stem={"k": 9, "stride": 5, "out": 256})


def test_engine_plan_sizes_and_addressing_limits(built):
    """chiron_engine_plan (no GPU): frame counts, the per-slot footprint of BASELINE configs[1], and the refusal of a
    max_batch whose activations / recurrent outputs pass the kernels' 32-bit addressing (0xFFFE0000 bytes per tensor)
    -- fp32, segment 400, 256 channels: 10485 windows fit, 10486 do not; halves double that."""
    from chiron_amd.engine import plan_sizes
    dna, rna = ca.dna_default_spec(), ca.rna_default_spec()
    p = plan_sizes(dna, 1100, 400, n_slots=3)
    assert p["T"] == 400 and p["ratio"] == 1.0 and p["tensor_limit_bytes"] == 0xFFFE0000
    assert p["largest_tensor_bytes"] == 1100 * 400 * 256 * 4
    z, lasth, act = 400 * 1104 * 800 * 4, 400 * 1104 * 200 * 4, 1100 * 400 * 256 * 4
    assert z + 2 * lasth + 3 * act < p["slot_bytes"] < z + 2 * lasth + 3 * act + 40e6 and p["total_bytes"] == 3 * p["slot_bytes"]
    assert plan_sizes(rna, 400, 500)["T"] == 100 and plan_sizes(ca.rna_head_spec("rna_model3"), 8, 500)["ratio"] == 500 / 72
    assert plan_sizes(dna, 1100, 400, max_beam=50)["slot_bytes"] > p["slot_bytes"] + 5e8
    assert plan_sizes(dna, 10485, 400)["largest_tensor_bytes"] <= 0xFFFE0000
    for kw in (dict(max_batch=10486, segment_len=400), dict(max_batch=30000, segment_len=400, dtype="fp16"),
               dict(max_batch=1100, segment_len=4000), dict(max_batch=6000000, segment_len=400)):
        with pytest.raises(_lib.ChironError) as ei:
            plan_sizes(dna, **kw)
        assert ei.value.status == _lib.ERR_OVERFLOW, kw
    assert plan_sizes(dna, 20000, 400, dtype="fp16")["T"] == 400          # halves: twice the rows
    assert plan_sizes(dna, 4096, 400, dtype="fp16-w2") == plan_sizes(dna, 4096, 400, dtype="fp16")   # the same buffers (z is fp32 in both)
    with pytest.raises(_lib.ChironError) as ei:
        plan_sizes(dna, 0, 400)
    assert ei.value.status == _lib.ERR_INVALID
    # the constructor path refuses before it looks for a GPU
    lib = _lib.load()
    d = dna.to_c()
    o = _lib.EngineOpts(0, 10486, 400, 1, _lib.F32, 0)
    blob = np.zeros(sum(int(np.prod(v)) for v in dna.variables().values()), np.float32)
    h = C.c_void_p()
    assert lib.chiron_engine_create(C.byref(d), blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(o), C.byref(h)) == _lib.ERR_OVERFLOW


def test_every_environment_switch_of_the_product_is_documented():
    """The engine's A/B and debugging switches are read with getenv("CHIRON_...") in csrc/ and os.environ in the Python
    host side; INTEGRATION.md's table is what a maintainer sees.  A switch missing from it fails here."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for path in glob.glob(os.path.join(root, "chiron_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "chiron_amd", "csrc", "*.cpp")):
        found.update(re.findall(r'getenv\("(CHIRON_[A-Z0-9_]+)"\)', open(path).read()))
    for path in glob.glob(os.path.join(root, "chiron_amd", "*.py")):
        found.update(re.findall(r'environ(?:\.get)?[\(\[]\s*"(CHIRON_[A-Z0-9_]+)"', open(path).read()))
    assert len(found) > 10            # the scan itself works
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(e for e in found if e not in doc)
    assert not missing, "environment switches not in INTEGRATION.md: %s" % missing


def test_the_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: only tests/, tools/ measurement scripts, __graft_entry__.smoke() and bench.py's cpu_baseline leg
    may import or load anything under it.  The package holds no import of it, no path into it, and bench.py's only use sits inside
    cpu_baseline()."""
    import ast
    pkg = os.path.join(ROOT, "chiron_amd")
    for dirpath, _, names in os.walk(pkg):
        for n in names:
            if n.endswith(".py"):
                tree = ast.parse(open(os.path.join(dirpath, n)).read())
                for node in ast.walk(tree):
                    mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""] if isinstance(node, ast.ImportFrom) else []
                    assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), (n, mods)
            if n.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, n)).read()
                assert "libchiron_oracle" not in text and "oracle/_build" not in text, n
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = set()
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef):
            for node in ast.walk(fn):
                if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                    users.add(fn.name)
                if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                    users.add(fn.name)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any((getattr(n, "module", "") or "").startswith("oracle") or any(a.name.startswith("oracle") for a in n.names) for n in top)
    assert users == {"cpu_baseline"}, users
