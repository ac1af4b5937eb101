#!/usr/bin/env python3
"""Pin the NN oracle to the graphs the reference ships.

Runs ONLY in the build container (needs /root/reference).  Reads
    chiron/model/DNA_default/final.ckpt-158301.meta   (MetaGraphDef, TF 1.13.1 producer; trained B=300, L=400)
    chiron/model/RNA_default/final.ckpt-80000.meta    (TF 1.8.0; B=100, L=2000)
and writes two kinds of fixture (data only; no reference source text, no protobuf bytes):

  tests/golden/meta_graph.json       a digest of what the graphs record: every Conv2D (filter variable, strides,
                                     padding, what feeds it), the BN arithmetic of both tf.cond branches with its
                                     constants, each residual block's composition, the dynamic_rnn while-loop bodies
                                     (LSTM cell expression, masking Selects, loop bound), where ReverseSequence sits,
                                     how the layers are concatenated, the FC head, and the attrs of the
                                     CTCBeamSearchDecoder / CTCLoss nodes the training graph carries.
  tests/golden/meta_golden_<m>.npz   activations obtained by EXECUTING the reference's node list (tf1_graph.GraphEval,
                                     float64) on seeded inputs at the graph's own static shape: the signal batch,
                                     seq_len (ragged, including 0 and 1), full logits of a row subset, per-row sums
                                     of all logits and of the CNN features.  DNA additionally with training=True,
                                     i.e. the batch-statistics branch of the BN tf.cond (what HEAD's simple_global_bn
                                     computes, cnn.py:166-188).

    python tests/golden/make_meta_golden.py           # ~5 min of numpy
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf1_graph as g  # noqa: E402

REF = "/root/reference/chiron/model"
MODELS = {
    "dna": ("DNA_default/final.ckpt-158301.meta", "dna_default_spec"),
    "rna": ("RNA_default/final.ckpt-80000.meta", "rna_default_spec"),
}
LOGITS = "rnn_fnn_layer/rnn_logits_rs"
WEIGHT_SEED = 11
ROWS_PER_MODEL = 12


def closure(nodes, name):
    seen, stack = set(), [name]
    while stack:
        x = stack.pop()
        if x in seen:
            continue
        seen.add(x)
        stack.extend(s for s, _ in nodes[x].inputs)
    return seen


def short(name, prefix):
    return name[len(prefix):] if name.startswith(prefix) else name


def weights_digest(w):
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k], dtype="<f4").tobytes())
    return h.hexdigest()


def digest(nodes):
    """The structural record (strings and small numbers only)."""
    order = {n: i for i, n in enumerate(nodes)}
    used = closure(nodes, LOGITS)
    inf = sorted(used, key=order.get)
    out = {}
    out["placeholders"] = [{"name": n, "dtype": g.DT_NAMES.get(nodes[n].attr["dtype"][1]), "shape": nodes[n].attr["shape"][1]}
                           for n in inf if nodes[n].op == "Placeholder"]
    out["variables"] = [{"name": n, "shape": nodes[n].attr["shape"][1]} for n in inf if nodes[n].op == "VariableV2"]

    # ---- BN sites: the tf.cond Merge of every <site>_bn; leaves make the expressions readable
    bn_merges = [n for n in inf if nodes[n].op == "Merge" and n.endswith("/cond/Merge")]
    leaves = {"Reshape": "signal[B,1,L,1]"}
    sites = []
    for m in bn_merges:
        site = m.split("_bn/")[0]
        leaves[m] = "BN(%s)" % site
        sites.append((site, m))
    convs = []
    for n in inf:
        if nodes[n].op != "Conv2D":
            continue
        fvar = nodes[nodes[n].inputs[1][0]].inputs[0][0]
        convs.append({"name": n, "filter": fvar, "filter_shape": nodes[fvar].attr["shape"][1],
                      "strides": nodes[n].attr["strides"], "padding": nodes[n].attr["padding"],
                      "data_format": nodes[n].attr.get("data_format"), "dilations": nodes[n].attr.get("dilations"),
                      "input": g.expression(nodes, nodes[n].inputs[0][0], nodes[n].inputs[0][1], leaves)})
    out["conv2d"] = convs
    bns = []
    for site, m in sites:
        pfx = m[:-len("Merge")]
        conv_out = nodes[pfx + "batchnorm_1/mul_1/Switch"].inputs[0][0]
        lv = {conv_out: "x"}
        for leaf in ("scale", "offset", "pop_mean", "pop_var"):
            lv[site + "_bn/" + leaf] = leaf
        s_inf, s_trn = g.Symbolic(nodes, lv, through_cond=True), g.Symbolic(nodes, lv, through_cond=True)
        bns.append({"site": site, "conv": conv_out,
                    "merge_inputs": [s for s, _ in nodes[m].inputs],
                    # the arithmetic of each tf.cond branch (Switch nodes printed as their data input) and the Switch
                    # output port its operands arrive through: 0 = pred false (inference), 1 = pred true (training)
                    "inference_branch": g.expression(nodes, pfx + "batchnorm_1/add_1", 0, sym=s_inf),
                    "inference_switch_ports": sorted(s_inf.cond_ports),
                    "training_branch": g.expression(nodes, pfx + "batchnorm/add_1", 0, sym=s_trn),
                    "training_switch_ports": sorted(s_trn.cond_ports),
                    "epsilon_inference": float(nodes[pfx + "batchnorm_1/add/y"].attr["value"]),
                    "epsilon_training": float(nodes[pfx + "batchnorm/add/y"].attr["value"]),
                    "pred": g.expression(nodes, pfx + "pred_id", 0, {})})
    out["batch_norm"] = bns

    # ---- residual blocks: relu(add(branch1, branch2)) outputs
    blocks = []
    for n in inf:
        if nodes[n].op == "Relu" and "/branch" not in n:
            blocks.append({"name": n, "expr": g.expression(nodes, n, 0, leaves)})
            leaves[n] = "OUT(%s)" % n.split("/")[0]
    out["residual_blocks"] = blocks
    out["cnn_feature"] = g.expression(nodes, "fea_rs", 0, leaves)
    leaves["fea_rs"] = "cnn_feature[B,T,C]"

    # ---- dynamic_rnn while loops
    frames = {}
    for n in inf:
        if nodes[n].op == "Enter":
            frames.setdefault(nodes[n].attr["frame_name"], []).append(n)
    loops = []
    gather_leaf = {}
    for fname in sorted(frames, key=lambda f: min(order[e] for e in frames[f])):
        pfx = fname[:-len("while_context")]          # ".../fw/fw/while/"
        scope = pfx[:-len("while/")]
        merges = [n for n in inf if nodes[n].op == "Merge" and n.startswith(pfx)]
        lv = dict(leaves)
        loopvars = []
        roots = {}
        for k, m in enumerate(merges):
            enter = [s for s, _ in nodes[m].inputs if nodes[s].op == "Enter"][0]
            nxt = [s for s, _ in nodes[m].inputs if nodes[s].op == "NextIteration"][0]
            lv[m] = "v%d" % k
            loopvars.append({"var": "v%d" % k, "merge": short(m, scope), "init": g.expression(nodes, enter, 0, leaves)})
            roots["next_v%d" % k] = nodes[nxt].inputs[0]
        for r in [n for n in inf if nodes[n].op == "TensorArrayReadV3" and n.startswith(pfx)]:
            lv[r] = "x_t"
        for n_ge in [n for n in inf if nodes[n].op == "GreaterEqual" and n.startswith(pfx)]:
            lv[nodes[n_ge].inputs[1][0]] = "seq_len"
        writes = [n for n in inf if nodes[n].op == "TensorArrayWriteV3" and n.startswith(pfx)]
        scat = [n for n in inf if nodes[n].op == "TensorArrayScatterV3" and n.startswith(scope)][0]
        gath = [n for n in inf if nodes[n].op == "TensorArrayGatherV3" and n.startswith(scope)][0]
        cond = [n for n in inf if nodes[n].op == "LoopCond" and n.startswith(pfx)][0]
        roots["loop_cond"] = (cond, 0)
        roots["output_written"] = nodes[writes[0]].inputs[2]
        roots["output_write_index"] = nodes[writes[0]].inputs[1]
        loops.append({
            "frame": fname, "loop_vars": loopvars,
            "body": g.Symbolic(nodes, lv).render(roots),
            "input_unstacked_from": g.expression(nodes, nodes[scat].inputs[2][0], nodes[scat].inputs[2][1], leaves),
            "output_element_shape": nodes[gath].attr["element_shape"][1],
        })
        gather_leaf[gath] = "LOOP_OUT(%s)" % scope.rstrip("/")
        leaves[gath] = gather_leaf[gath]
        # expose this loop's stacked output to later loops / the head under a short name
    out["rnn_loops"] = loops
    lasth = nodes["rnn_fnn_layer/lasth_rs"].inputs[0][0]
    out["rnn_output"] = g.expression(nodes, lasth, 0, leaves)
    # per-layer concat nodes (stack_bidirectional: one per layer; bidirectional(MultiRNN): one at the end)
    out["rnn_concats"] = [{"name": n, "expr": g.expression(nodes, n, 0, leaves)} for n in inf
                          if nodes[n].op == "ConcatV2" and nodes[n].name.split("/")[-1] == "concat"
                          and nodes[n].name.count("/") <= 3 and "rnn" in n.lower() and "while" not in n]
    out["reverse_sequence"] = [{"name": n, "seq_dim": nodes[n].attr["seq_dim"], "batch_dim": nodes[n].attr.get("batch_dim", 0),
                                "input": g.expression(nodes, nodes[n].inputs[0][0], nodes[n].inputs[0][1], leaves),
                                "lengths": g.expression(nodes, nodes[n].inputs[1][0], 0, leaves)}
                               for n in inf if nodes[n].op == "ReverseSequence"]
    out["fc_head"] = g.expression(nodes, LOGITS, 0, {lasth: "lasth[B,T,2H]"})
    out["ctc_nodes"] = [{"name": n.name, "op": n.op, "attrs": {k: v for k, v in n.attr.items() if not k.startswith("_") and k != "T"},
                         "inputs": [g.expression(nodes, s, i, {LOGITS: "logits[B,T,5]"}) for s, i in n.inputs]}
                        for n in nodes.values() if n.op in ("CTCBeamSearchDecoder", "CTCGreedyDecoder", "CTCLoss")]
    return out


def seeded_case(model, spec, batch, length):
    """Deterministic inputs (generated by the product's seeded helpers; their digests are stored so drift is detected)."""
    import chiron_amd as ca
    w = ca.synthetic_weights(spec, seed=WEIGHT_SEED)
    x = np.stack([ca.synthetic_signal(1, length, seed=1000 + 7 * i)[0] for i in range(batch)]).astype(np.float32)
    T = spec.output_len(length)
    rng = np.random.RandomState(5 if model == "dna" else 6)
    seq = np.full(batch, T, np.int32)
    ragged = rng.choice(batch, batch // 3, replace=False)
    seq[ragged] = rng.randint(2, T, len(ragged))
    seq[3], seq[4], seq[5], seq[7] = 0, 1, T - 1, T // 3
    # trailing samples of short rows are zero padding, as read_data_for_eval produces (chiron_input.py:253-292)
    ratio = length / T
    for b in range(batch):
        x[b, int(round(seq[b] * ratio)):] = 0
    return w, x, seq


def run_model(model, meta_rel, spec_fn):
    import chiron_amd as ca
    t0 = time.time()
    nodes = g.load_meta_graph(os.path.join(REF, meta_rel))
    dg = digest(nodes)
    ph = {p["dtype"] + str(len(p["shape"])): p for p in dg["placeholders"]}
    x_ph, seq_ph, train_ph = ph["float322"], ph["int321"], ph["bool0"]
    batch, length = x_ph["shape"]
    spec = getattr(ca, spec_fn)()
    w, x, seq = seeded_case(model, spec, batch, length)
    missing = [v["name"] for v in dg["variables"] if v["name"] not in w]
    assert not missing, missing
    for v in dg["variables"]:
        assert list(w[v["name"]].shape) == v["shape"], v
    rng = np.random.RandomState(99)
    rows = sorted(set([0, 1, 3, 4, 5, 7, batch - 1]) | set(rng.choice(batch, ROWS_PER_MODEL, replace=False).tolist()))[:ROWS_PER_MODEL + 4]
    rows = np.array(rows, np.int32)
    arrays = {"x": x.astype(np.int16), "seq_len": seq, "rows": rows}
    assert np.array_equal(arrays["x"].astype(np.float32), x)
    info = {"batch": batch, "segment_len": length, "weight_seed": WEIGHT_SEED, "weights_sha256": weights_digest(w),
            "x_placeholder": x_ph["name"], "seq_len_placeholder": seq_ph["name"], "training_placeholder": train_ph["name"]}
    modes = [("population", False)] + ([("batch", True)] if model == "dna" else [])
    for mode, training in modes:
        feeds = dict(w)
        feeds[train_ph["name"]] = np.bool_(training)
        feeds[x_ph["name"]] = x
        feeds[seq_ph["name"]] = seq
        ev = g.GraphEval(nodes, feeds, np.float64)
        fea = ev.run("fea_rs")
        logits = ev.run(LOGITS)
        assert logits.shape == (batch, fea.shape[1], 5) and np.isfinite(logits).all()
        arrays["logits_rows_" + mode] = logits[rows]
        arrays["logits_rowsum_" + mode] = logits.sum(axis=(1, 2))
        arrays["logits_abssum_" + mode] = np.abs(logits).sum(axis=(1, 2))
        arrays["fea_rowsum_" + mode] = fea.sum(axis=(1, 2))
        arrays["fea_rows_" + mode] = fea[rows][:, ::max(1, fea.shape[1] // 8)]      # 8-9 frames of every golden row
        info["loop_iterations_" + mode] = {k: int(v[1]) for k, v in ev.frame_results.items()}
        info["ops_executed_" + mode] = dict(sorted(ev.op_counts.items()))
        print("%s %s: logits %s max|.| %.3f  (%.0f s)" % (model, mode, logits.shape, np.abs(logits).max(), time.time() - t0))
    info["T"] = int(arrays["logits_rows_population"].shape[1])
    np.savez_compressed(os.path.join(HERE, "meta_golden_%s.npz" % model), **arrays)
    dg["golden"] = info
    return dg


def main():
    out = {"_generated_by": "tests/golden/make_meta_golden.py",
           "_source": {m: "chiron/model/" + rel for m, (rel, _) in MODELS.items()}}
    path = os.path.join(HERE, "meta_graph.json")
    digest_only = "--digest-only" in sys.argv      # re-render the structural record, keep the evaluated activations
    old = json.load(open(path)) if digest_only else None
    for model, (rel, spec_fn) in MODELS.items():
        if digest_only:
            out[model] = digest(g.load_meta_graph(os.path.join(REF, rel)))
            out[model]["golden"] = old[model]["golden"]
        else:
            out[model] = run_model(model, rel, spec_fn)
    with open(os.path.join(HERE, "meta_graph.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
