"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/chiron_amd.h
declares; argument validation and the host-only entry point (chiron_assemble) work without a GPU.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import chiron_amd as ca
from chiron_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "chiron_amd.h")).read()
    declared = set(re.findall(r"\b(chiron_[a-z_]+)\s*\(", header))
    declared -= {"chiron_status"}
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, "binding and header disagree: %s" % (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.chiron_abi_version() == 1


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
    st = lib.chiron_assemble(bases.ctypes.data, off.ctypes.data, 2, None, 7, None, None, 0, C.byref(n))
    assert st == _lib.ERR_INVALID
    counts = np.zeros((4, 1))
    st = lib.chiron_assemble(bases.ctypes.data, off.ctypes.data, 2, None, _lib.KERNAL_STICK, counts.ctypes.data, None, 1, C.byref(n))
    assert st == _lib.ERR_OVERFLOW and n.value == 4
