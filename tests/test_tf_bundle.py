"""CPU: TensorFlow checkpoint bundle reader (row W of SURVEY.md 8a): parses the shipped .index files
(names, shapes, offsets pinned by SURVEY appendix B) and round-trips synthetic weights through a
bundle written by tests/bundle_writer.py."""
import json
import os
import shutil

import numpy as np
import pytest

import chiron_amd as ca
from chiron_amd import tf_bundle
from bundle_writer import write_bundle, crc32c

PKG_MODELS = os.path.join(os.path.dirname(os.path.abspath(ca.__file__)), "model")


@pytest.mark.parametrize("name,prefix,data_bytes,spec_fn", [
    ("DNA_default", "final.ckpt-158301", 21887048, ca.dna_default_spec),
    ("RNA_default", "final.ckpt-80000", 27831368, ca.rna_default_spec)])
def test_shipped_index_decodes_to_the_expected_topology(name, prefix, data_bytes, spec_fn):
    d = os.path.join(PKG_MODELS, name)
    assert tf_bundle.latest_checkpoint(d) == os.path.join(d, prefix)
    entries = tf_bundle.read_index(os.path.join(d, prefix + ".index"))
    assert len(entries) == 167
    infer = {k: v for k, v in entries.items() if "/Adam" not in k and not k.endswith("_power") and k != "global_step"}
    assert len(infer) == 68
    assert max(v["offset"] + v["size"] for v in entries.values()) == data_bytes
    assert entries["global_step"]["dtype"] == tf_bundle.DT_INT32
    spec = ca.spec_from_variables({k: v["shape"] for k, v in entries.items()})
    want = spec_fn()
    assert spec.blocks == want.blocks and spec.rnn_kind == want.rnn_kind and spec.bn_mode == "population"
    for k, shape in want.variables().items():
        assert tuple(entries[k]["shape"]) == tuple(shape) and entries[k]["dtype"] == tf_bundle.DT_FLOAT
        assert entries[k]["size"] == 4 * int(np.prod(shape))
    # the shipped folders have no data blob: load_model must say so, or fall back when allowed
    with pytest.raises(FileNotFoundError):
        ca.load_model(d)
    s2, w2, cfg = ca.load_model(d, allow_synthetic=True)
    assert s2.blocks == want.blocks and set(w2) == set(want.variables())
    assert cfg["rnn"]["hidden_num"] == 100


def test_crc32c_known_answer():
    assert crc32c(b"123456789") == 0xE3069283


@pytest.mark.parametrize("spec_fn", [ca.dna_default_spec, ca.rna_default_spec])
def test_round_trip_through_a_written_bundle(tmp_path, spec_fn):
    spec = spec_fn()
    w = ca.synthetic_weights(spec, seed=5)
    d = str(tmp_path)
    tensors = dict(w)
    for k in list(w)[:3]:                          # optimiser slots must be skipped by the loader
        tensors[k + "/Adam"] = np.zeros_like(w[k])
    write_bundle(os.path.join(d, "final.ckpt-7"), tensors, extra_int32={"global_step": 7})
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "final.ckpt-7"\n')
    json.dump({"cnn": {"model": "dna_model1"}, "rnn": {"layer_num": 3, "hidden_num": 100, "cell_type": "LSTM",
                                                         "layer_type": "normal"}}, open(os.path.join(d, "model.json"), "w"))
    spec2, w2, _ = ca.load_model(d)
    assert spec2.blocks == spec.blocks and spec2.rnn_kind == spec.rnn_kind
    assert list(w2) == list(spec.variables())
    for k in w:
        assert np.array_equal(w2[k], w[k])
    assert np.array_equal(spec2.pack(w2), spec.pack(w))
    # truncated data file -> loud error
    data = os.path.join(d, "final.ckpt-7.data-00000-of-00001")
    open(data, "r+b").truncate(os.path.getsize(data) // 2)
    with pytest.raises(IOError):
        ca.load_model(d)
