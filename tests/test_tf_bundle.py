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


def test_reader_verifies_the_stored_crc32c(tmp_path):
    """BundleEntryProto.crc32c: one flipped byte in the data file is reported by variable name;
    verify=False still reads it (forensics); the reader's CRC equals the writer's independent bytewise one."""
    from chiron_amd import tf_bundle
    rng = np.random.RandomState(2)
    for n in (0, 1, 7, 8, 9, 4096 + 3):
        raw = rng.bytes(n)
        assert tf_bundle.crc32c(raw) == crc32c(raw)
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283
    spec = ca.dna_default_spec()
    w = ca.synthetic_weights(spec, seed=6)
    prefix = os.path.join(str(tmp_path), "final.ckpt-1")
    write_bundle(prefix, dict(w))
    entries = tf_bundle.read_index(prefix + ".index")
    victim = "res_layer2/branch2/conv2b/weights"
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(entries[victim]["offset"] + 1234)
        b = f.read(1)
        f.seek(-1, 1)
        f.write(bytes([b[0] ^ 0x40]))
    with pytest.raises(IOError, match="checksum mismatch in variable 'res_layer2/branch2/conv2b/weights'"):
        tf_bundle.read_tensors(prefix, entries, list(spec.variables()))
    got = tf_bundle.read_tensors(prefix, entries, [victim], verify=False)[victim]
    assert (got != w[victim]).sum() == 1
    ok = tf_bundle.read_tensors(prefix, entries, [n for n in spec.variables() if n != victim])
    assert all(np.array_equal(ok[k], w[k]) for k in ok)


def test_head_style_batch_bn_checkpoint_loads(tmp_path):
    """A model trained with HEAD's simple_global_bn (cnn.py:166-188) stores <site>_bn/<leaf>_bn_scale|_bn_offset and no
    statistics (last paragraph).  load_model must recognise it (bn_mode = batch), read exactly those
    names, and hand the engine a blob whose unused pop_mean / pop_var slots are 0 / 1."""
    spec_b = ca.dna_default_spec(bn_mode="batch")
    w = ca.synthetic_weights(ca.dna_default_spec(), seed=8)          # canonical names
    tensors = {}
    for name, a in w.items():
        if name.endswith("_bn/pop_mean") or name.endswith("_bn/pop_var"):
            continue
        if name.endswith("_bn/scale") or name.endswith("_bn/offset"):
            site, leaf = name.rsplit("_bn/", 1)
            name = "%s_bn/%s_bn_%s" % (site, site.split("/")[-1], leaf)
        tensors[name] = a
    assert set(tensors) == set(spec_b.variables()) and len(tensors) == 68 - 20
    d = str(tmp_path)
    write_bundle(os.path.join(d, "model.ckpt-3"), tensors, extra_int32={"global_step": 3})
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-3"\n')
    json.dump({"cnn": {"model": "dna_model1"}, "rnn": {"layer_num": 3, "hidden_num": 100, "cell_type": "LSTM",
                                                         "layer_type": "normal"}}, open(os.path.join(d, "model.json"), "w"))
    spec2, w2, _ = ca.load_model(d)
    assert spec2.bn_mode == "batch" and spec2.blocks == spec_b.blocks
    assert list(w2) == list(spec_b.blob_layout()) == list(ca.dna_default_spec().variables())
    site = "res_layer2/branch2/conv2b"
    assert np.array_equal(w2[site + "_bn/scale"], w[site + "_bn/scale"]) and np.array_equal(w2[site + "_bn/offset"], w[site + "_bn/offset"])
    assert np.all(w2[site + "_bn/pop_mean"] == 0) and np.all(w2[site + "_bn/pop_var"] == 1)
    blob = spec2.pack(w2)
    assert blob.size == spec_b.pack(w).size
    # the oracle consumes the same canonical dict (batch statistics ignore the filled slots)
    from oracle import nn_oracle
    x = ca.synthetic_signal(1, 2 * 64, seed=1)[0].reshape(2, 64)
    a, _ = nn_oracle.inference(x, [64, 64], spec2.to_dict(), w2, dtype=np.float64)
    b, _ = nn_oracle.inference(x, [64, 64], spec_b.to_dict(), w, dtype=np.float64)
    assert np.array_equal(a, b)
    # HEAD names handed straight to pack() work too; a missing scale is reported by name
    assert np.array_equal(spec_b.pack(tensors), blob)
    del tensors["res_layer1/branch1/conv1_bn/conv1_bn_scale"]
    with pytest.raises(KeyError, match="conv1_bn_scale"):
        spec_b.pack(tensors)
