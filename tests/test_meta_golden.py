"""The NN oracle pinned to the graphs the reference ships.

tests/golden/meta_graph.json and meta_golden_{dna,rna}.npz are produced by tests/golden/make_meta_golden.py from
chiron/model/{DNA,RNA}_default/final.ckpt-*.meta: a structural digest of the MetaGraphDef, and activations obtained by
executing the reference's own node list (tf1_graph.GraphEval, float64) on seeded inputs at the graph's static shape
(DNA 300 x 400, RNA 100 x 2000; ragged seq_len including 0 and 1).  Here:

  * CPU: the facts the product and the oracle hard-code (strides, kernel widths, SAME padding, BN epsilon and
    association order, LSTM gate order / forget bias / masking, ReverseSequence placement, layer concatenation, FC head,
    decoder attrs) are read back from the digest;
  * CPU: oracle/nn_oracle.py (float64) reproduces the executed graph to 1e-12 -- population BN for both models and the
    batch-statistics branch (training=True, what HEAD's simple_global_bn computes) for DNA;
  * GPU (-m gpu): the HIP engine's logits against the same fixture at the north-star tolerance 1e-4.
"""
import hashlib
import json
import os
import re

import numpy as np
import pytest

import chiron_amd as ca
from oracle import nn_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLDEN, "meta_graph.json")) as f:
        return json.load(f)


def _case(meta, model):
    z = np.load(os.path.join(GOLDEN, "meta_golden_%s.npz" % model))
    spec = ca.dna_default_spec() if model == "dna" else ca.rna_default_spec()
    info = meta[model]["golden"]
    w = ca.synthetic_weights(spec, seed=info["weight_seed"])
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k], dtype="<f4").tobytes())
    assert h.hexdigest() == info["weights_sha256"], \
        "synthetic_weights() no longer produces the weights the fixture was generated with: re-run tests/golden/make_meta_golden.py"
    return spec, w, z, info


# ------------------------------------------------------------------------------------------ digest: structural facts
def test_conv_sites_match_the_specs(meta):
    for model, spec in (("dna", ca.dna_default_spec()), ("rna", ca.rna_default_spec())):
        convs = {c["name"].rsplit("/", 1)[0]: c for c in meta[model]["conv2d"]}
        assert len(convs) == 4 * len(spec.blocks)
        for b in spec.blocks:
            n = b["name"]
            for site, k, cin, stride in ((n + "/branch1/conv1", 1, b["in"], b["stride"]),
                                         (n + "/branch2/conv2a", 1, b["in"], 1),
                                         (n + "/branch2/conv2b", b["k"], b["out"], b["stride"]),
                                         (n + "/branch2/conv2c", 1, b["out"], 1)):
                c = convs[site]
                assert c["filter"] == site + "/weights"
                assert c["filter_shape"] == [1, k, cin, b["out"]]
                assert c["strides"] == [1, 1, stride, 1], (site, c["strides"])
                assert c["padding"] == "SAME" and c["data_format"] == "NHWC" and c["dilations"] == [1, 1, 1, 1]
        # the variable manifest of the graph == the blob contract of include/chiron_amd.h
        assert {v["name"]: tuple(v["shape"]) for v in meta[model]["variables"]} == dict(spec.variables())
        # topology derived from names/shapes alone (what load_model does for a checkpoint) agrees, strides included
        derived = ca.model.spec_from_variables({v["name"]: v["shape"] for v in meta[model]["variables"]},
                                               strides={c["name"].split("/")[0]: c["strides"][2] for c in meta[model]["conv2d"]
                                                        if "/conv2b/" in c["name"]})
        assert derived.to_dict() == spec.to_dict()
        default = ca.model.spec_from_variables({v["name"]: v["shape"] for v in meta[model]["variables"]})
        assert default.to_dict() == spec.to_dict()


def test_residual_block_composition(meta):
    for model in ("dna", "rna"):
        blocks = meta[model]["residual_blocks"]
        assert [b["name"].split("/")[0] for b in blocks] == ["res_layer1", "res_layer2", "res_layer3"]
        # res_layer1: BN on the shortcut (i_bn); res_layer2/3: bare 1x1 conv shortcut (cnn.py:380-389)
        assert blocks[0]["expr"] == "Relu(Add(BN(res_layer1/branch1/conv1), BN(res_layer1/branch2/conv2c)))"
        for k in (2, 3):
            assert re.fullmatch(r"Relu\(Add\(Conv2D\{strides=\[1, 1, 1, 1\],padding=SAME\}\(OUT\(res_layer%d\), "
                                r"var\(res_layer%d/branch1/conv1/weights\)\), BN\(res_layer%d/branch2/conv2c\)\)\)" % (k - 1, k, k),
                                blocks[k - 1]["expr"])
        convs = {c["name"].rsplit("/", 1)[0]: c["input"] for c in meta[model]["conv2d"]}
        assert convs["res_layer1/branch1/conv1"] == "signal[B,1,L,1]" and convs["res_layer1/branch2/conv2a"] == "signal[B,1,L,1]"
        for n in ("res_layer1", "res_layer2", "res_layer3"):
            assert convs[n + "/branch2/conv2b"] == "Relu(BN(%s/branch2/conv2a))" % n      # BN then ReLU
            assert convs[n + "/branch2/conv2c"] == "Relu(BN(%s/branch2/conv2b))" % n
        assert meta[model]["cnn_feature"].startswith("Reshape(OUT(res_layer3)")


def test_batch_norm_arithmetic(meta):
    eps32 = float(np.float32(1e-5))
    assert nn_oracle.BN_EPS == eps32
    for model in ("dna", "rna"):
        assert len(meta[model]["batch_norm"]) == 10
        for bn in meta[model]["batch_norm"]:
            assert bn["epsilon_inference"] == eps32 and bn["epsilon_training"] == eps32
            assert bn["pred"] == "placeholder(Placeholder)"          # the `training` bool
            # inference (pred false, Switch port 0): inv = rsqrt(pop_var + eps) * scale ; y = x*inv + (offset - pop_mean*inv)
            assert bn["inference_switch_ports"] == [0] and bn["training_switch_ports"] == [1]
            assert bn["inference_branch"] == ("Add(Mul(x, t1), Sub(offset, Mul(pop_mean, t1))) where "
                                              "t1 = Mul(Rsqrt(Add(pop_var, %r)), scale)" % eps32)
            # training (port 1): moments over axes [0,1,2] of this batch, biased variance, same association order
            assert bn["training_branch"] == (
                "Add(Mul(x, t2), Sub(offset, Mul(Squeeze(t1), t2))) where t1 = Mean{keep_dims=True}(x, [0, 1, 2]); "
                "t2 = Mul(Rsqrt(Add(Squeeze(Mean{keep_dims=True}(SquaredDifference(x, t1), [0, 1, 2])), %r)), scale)" % eps32)


def _loop_lets(loop):
    return dict(loop["body"]["let"]), loop["body"]


def test_lstm_loop_bodies(meta):
    """TF LSTMCell under dynamic_rnn(sequence_length): gate order i, j, f, o; forget bias 1.0 added before the sigmoid;
    frames t >= seq_len emit zeros and carry (c, h); loop bound min(T, max(1, max seq_len))."""
    for model, n_loops, cells in (("dna", 6, 1), ("rna", 2, 3)):
        loops = meta[model]["rnn_loops"]
        assert len(loops) == n_loops
        for loop in loops:
            lets, body = _loop_lets(loop)
            text = json.dumps(loop["body"])
            splits = [v for v in lets.values() if v.startswith("Split{num_split=4}(1, BiasAdd(MatMul(ConcatV2(")]
            assert len(splits) == cells
            for sp in splits:      # z = [x, h] @ kernel + bias, split in 4 along axis 1
                assert re.search(r"MatMul\(ConcatV2\((x_t|t\d+), v\d+, 1\), var\([^)]*lstm_cell/kernel\)\), var\([^)]*lstm_cell/bias\)\)", sp)
            cs = [v for v in lets.values() if re.fullmatch(
                r"Add\(Mul\(Sigmoid\(Add\((t\d+):2, 1\.0\)\), v\d+\), Mul\(Sigmoid\(\1:0\), Tanh\(\1:1\)\)\)", v)]
            assert len(cs) == cells, text           # c = sigmoid(f + 1.0) * c_prev + sigmoid(i) * tanh(j)
            hs = [v for v in lets.values() if re.fullmatch(r"Mul\(Sigmoid\(t\d+:3\), Tanh\(t\d+\)\)", v)]
            assert len(hs) == cells, text           # h = sigmoid(o) * tanh(c)
            assert "GreaterEqual(v1, seq_len)" in lets.values() or "GreaterEqual(v1, seq_len)" in text
            assert body["output_write_index"] == "v1" and body["next_v1"] == "Add(v1, 1)"
            mask = [k for k, v in lets.items() if v == "GreaterEqual(v1, seq_len)"][0]
            # output: zeros where masked; state: carried where masked
            out_expr = lets.get(body["output_written"], body["output_written"])
            assert re.fullmatch(r"Select\(%s, Fill\(ConcatV2\(\[\d+\], \[100\], 0\), 0\.0\), t\d+\)" % mask, out_expr)
            carried = [v for k, v in body.items() if k.startswith("next_v") and v.startswith("Select(%s, v" % mask)]
            assert len(carried) == 2 * cells
            for v in carried:
                assert re.fullmatch(r"Select\(%s, (v\d+), t\d+\)" % mask, v)
            assert re.fullmatch(r"LogicalAnd\(Less\(v0, (t\d+)\), Less\(v1, Minimum\(\1, Maximum\(1, "
                                r"Max\{keep_dims=False\}\(placeholder\(Placeholder_2\), \[0\]\)\)\)\)\)", body["loop_cond"])
            for lv in loop["loop_vars"][3:]:
                assert re.fullmatch(r"Fill\(ConcatV2\(\[\d+\], \[100\], 0\), 0\.0\)", lv["init"])    # zero initial (c, h)


def test_direction_and_layer_composition(meta):
    tr = "ConcatV2([1, 0], Range(2, 3, 1), 0)"          # perm [1, 0, 2]: time-major inside the loops
    dna = meta["dna"]
    src = "cnn_feature[B,T,C]"
    for layer in range(3):
        fw, bw = dna["rnn_loops"][2 * layer], dna["rnn_loops"][2 * layer + 1]
        assert "/cell_%d/" % layer in fw["frame"] and "/fw/fw/" in fw["frame"] and "/bw/bw/" in bw["frame"]
        assert fw["input_unstacked_from"] == "Transpose(%s, %s)" % (src, tr)
        assert bw["input_unstacked_from"] == "Transpose(ReverseSequence{seq_dim=1,batch_dim=0}(%s, placeholder(Placeholder_2)), %s)" % (src, tr)
        pre = "BDLSTM_rnn/BDLSTM_rnn/cell_%d" % layer
        cat = ("ConcatV2(Transpose(LOOP_OUT(%s/bidirectional_rnn/fw/fw), %s), ReverseSequence{seq_dim=1,batch_dim=0}("
               "Transpose(LOOP_OUT(%s/bidirectional_rnn/bw/bw), %s), placeholder(Placeholder_2)), 2)" % (pre, tr, pre, tr))
        assert dna["rnn_concats"][layer] == {"name": pre + "/concat", "expr": cat}
        src = cat                                           # stacked: layer l+1 consumes concat(fw, bw) of layer l
    assert dna["rnn_output"] == src
    # RNA: two loops of three cells each (MultiRNNCell), one concat at the end (rnn.py:140-145)
    rna = meta["rna"]
    fw, bw = rna["rnn_loops"]
    assert fw["input_unstacked_from"] == "Transpose(cnn_feature[B,T,C], %s)" % tr
    assert bw["input_unstacked_from"].startswith("Transpose(ReverseSequence{seq_dim=1,batch_dim=0}(cnn_feature[B,T,C]")
    assert re.fullmatch(r"ConcatV2\(Transpose\(LOOP_OUT\(BDGRU_rnn/BDGRU_rnn/fw/fw\), .*\), ReverseSequence\{seq_dim=1,batch_dim=0\}\("
                        r"Transpose\(LOOP_OUT\(BDGRU_rnn/BDGRU_rnn/bw/bw\), .*\), placeholder\(Placeholder_2\)\), 2\)", rna["rnn_output"])
    kernels = re.findall(r"var\((BDGRU_rnn/fw/multi_rnn_cell/cell_\d)/lstm_cell/kernel\)", json.dumps(fw["body"]))
    assert sorted(set(kernels)) == ["BDGRU_rnn/fw/multi_rnn_cell/cell_%d" % i for i in range(3)]
    for model in ("dna", "rna"):
        for rs in meta[model]["reverse_sequence"]:
            assert rs["seq_dim"] == 1 and rs["batch_dim"] == 0 and rs["lengths"] == "placeholder(Placeholder_2)"


def test_fc_head_and_decoder_attrs(meta):
    for model in ("dna", "rna"):
        fc = meta[model]["fc_head"]
        # [B,T,2,H] * weights[2,H] -> sum over the 2 -> + bias -> [B*T,H] @ weights_class -> + bias_class; no activation
        assert re.fullmatch(r"Reshape\(BiasAdd\(MatMul\(Reshape\(BiasAdd\(Sum\{keep_dims=False\}\(Mul\(Reshape\(lasth\[B,T,2H\], "
                            r"Pack\(t1, 400, 2, 100\)\), var\(rnn_fnn_layer/weights\)\), 2\), var\(rnn_fnn_layer/bias\)\), "
                            r"Pack\(Mul\(t1, 400\), 100\)\), var\(rnn_fnn_layer/weights_class\)\), var\(rnn_fnn_layer/bias_class\)\), "
                            r"Pack\(t1, 400, 5\)\) where t1 = .*", fc), fc
        ctc = {c["op"]: c for c in meta[model]["ctc_nodes"]}
        beam = ctc["CTCBeamSearchDecoder"]
        assert beam["attrs"] == {"beam_width": 30, "merge_repeated": False, "top_paths": 1}     # chiron_model.py:117-122
        assert beam["inputs"] == ["Transpose(logits[B,T,5], [1, 0, 2])", "placeholder(Placeholder_2)"]   # time-major, seq_len
        assert ctc["CTCLoss"]["attrs"]["ctc_merge_repeated"] is True


# --------------------------------------------------------------------- executed graph vs the oracle (float64, 1e-12)
@pytest.mark.parametrize("model", ["dna", "rna"])
def test_oracle_reproduces_the_executed_graph_population_bn(meta, model):
    spec, w, z, info = _case(meta, model)
    rows = z["rows"]
    x = z["x"].astype(np.float32)
    assert x.shape == (info["batch"], info["segment_len"])
    ref, ratio = nn_oracle.inference(x[rows], z["seq_len"][rows], spec.to_dict(), w, dtype=np.float64)
    assert ref.shape == z["logits_rows_population"].shape and ratio == info["segment_len"] / info["T"]
    assert np.abs(ref - z["logits_rows_population"]).max() < 1e-12
    fea = nn_oracle.cnn_forward(x[rows].astype(np.float64), spec.to_dict(), {k: v.astype(np.float64) for k, v in w.items()})
    step = max(1, fea.shape[1] // 8)
    assert np.abs(fea[:, ::step] - z["fea_rows_population"]).max() < 1e-12
    # the golden rows are a faithful sample: their sums are in the all-row table
    assert np.allclose(z["logits_rows_population"].sum(axis=(1, 2)), z["logits_rowsum_population"][rows], rtol=0, atol=1e-9)
    # loop trip counts: min(T, max(1, max seq_len)) -- here some row is full length
    assert set(info["loop_iterations_population"].values()) == {info["T"]}


def test_oracle_reproduces_the_executed_graph_batch_statistics(meta):
    """training=True: the BN tf.cond takes the moments-of-this-batch branch; rows interact, so all 300 are evaluated."""
    spec, w, z, info = _case(meta, "dna")
    spec_b = ca.dna_default_spec(bn_mode="batch")
    x = z["x"].astype(np.float32)
    ref, _ = nn_oracle.inference(x, z["seq_len"], spec_b.to_dict(), w, dtype=np.float64)
    assert np.abs(ref[z["rows"]] - z["logits_rows_batch"]).max() < 1e-11
    assert np.abs(ref.sum(axis=(1, 2)) - z["logits_rowsum_batch"]).max() < 1e-8
    assert np.abs(z["logits_rows_batch"] - z["logits_rows_population"]).max() > 1e-2      # it really is another function


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("model", ["dna", "rna"])
def test_engine_logits_match_the_executed_graph(meta, model):
    """HIP engine at the graph's own shape (DNA 300 x 400, RNA 100 x 2000) against activations of the reference's node
    list: every golden row within 1e-4 (north-star fp32 tolerance); every row's sum of T x 5 logits within 2e-2, i.e. a
    mean deviation of 1e-5 per logit (the sums catch a wrong row anywhere in the batch, not rounding)."""
    spec, w, z, info = _case(meta, model)
    x = z["x"].astype(np.float32)
    sl = z["seq_len"].astype(np.int32)
    with ca.Engine(spec, w, max_batch=info["batch"], segment_len=info["segment_len"]) as eng:
        assert eng.T == info["T"]
        res = eng.infer(x, sl, want_logits=True)
    lg = res.logits.astype(np.float64)
    err = np.abs(lg[z["rows"]] - z["logits_rows_population"]).max()
    assert err < 1e-4, err
    assert np.abs(lg.sum(axis=(1, 2)) - z["logits_rowsum_population"]).max() < 2e-2
    assert np.abs(np.abs(lg).sum(axis=(1, 2)) - z["logits_abssum_population"]).max() < 2e-2


@pytest.mark.gpu
def test_engine_batch_statistics_match_the_executed_graph(meta):
    spec, w, z, info = _case(meta, "dna")
    spec_b = ca.dna_default_spec(bn_mode="batch")
    with ca.Engine(spec_b, w, max_batch=info["batch"], segment_len=info["segment_len"]) as eng:
        res = eng.infer(z["x"].astype(np.float32), z["seq_len"].astype(np.int32), want_logits=True)
    lg = res.logits.astype(np.float64)
    assert np.abs(lg[z["rows"]] - z["logits_rows_batch"]).max() < 1e-4
    assert np.abs(lg.sum(axis=(1, 2)) - z["logits_rowsum_batch"]).max() < 2e-2
