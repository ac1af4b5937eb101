"""Model topology + weight contract (counterpart of chiron/chiron_model.py
read_config :37-48 and of the variable layout of the shipped checkpoints,
SURVEY.md appendix B).

The topology is DATA-DRIVEN: it is derived from the checkpoint's variable
names/shapes (`spec_from_variables`), not hard-coded from HEAD, because HEAD
and the shipped checkpoints disagree.
"""
import json
import math
import os
from collections import OrderedDict

import numpy as np

from . import _lib


class ModelSpec(object):
    """blocks: list of dicts {name,in,out,k,stride,i_bn}; rnn_kind 'stack'|'multi';
    stem: None or {k, stride, out} -- the conv_layer/conv1 + BN + ReLU in front of the blocks in HEAD's
    RNA_model2 / RNA_model3 (cnn.py:454-476)."""

    STEM_SITE = "conv_layer/conv1"

    def __init__(self, blocks, rnn_kind="stack", rnn_layers=3, hidden=100, classes=5, bn_mode="population", stem=None):
        self.blocks = [dict(b) for b in blocks]
        self.stem = dict(stem) if stem else None
        if self.stem and self.blocks[0]["in"] != self.stem["out"]:
            raise ValueError("first block takes %d channels, the stem produces %d" % (self.blocks[0]["in"], self.stem["out"]))
        self.rnn_kind = rnn_kind
        self.rnn_layers = int(rnn_layers)
        self.hidden = int(hidden)
        self.classes = int(classes)
        self.bn_mode = bn_mode
        if rnn_kind not in ("stack", "multi"):
            raise ValueError("Cell layer type unrecognized: %r" % (rnn_kind,))
        if bn_mode not in ("population", "batch"):
            raise ValueError("bn_mode must be 'population' or 'batch'")

    # -- variable layout -----------------------------------------------------
    def lstm_in_width(self, layer):
        if layer == 0:
            return self.blocks[-1]["out"]
        return 2 * self.hidden if self.rnn_kind == "stack" else self.hidden

    def lstm_scope(self, layer, direction):
        """rnn.py:63 (BDLSTM_rnn, stack) / rnn.py:146 (BDGRU_rnn, multi) variable scopes."""
        if self.rnn_kind == "stack":
            return "BDLSTM_rnn/cell_%d/bidirectional_rnn/%s/lstm_cell/" % (layer, direction)
        return "BDGRU_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell/" % (direction, layer)

    BN_LEAVES = ("scale", "offset", "pop_mean", "pop_var")

    def bn_names(self, site):
        """Checkpoint names of one BN site, in the order (scale, offset, pop_mean, pop_var); None = not stored."""
        if self.bn_mode == "population":      # cnn.py:125-163 naming (shipped checkpoints)
            return [site + "_bn/" + leaf for leaf in self.BN_LEAVES]
        leaf = site.split("/")[-1]            # cnn.py:181-186 naming (HEAD simple_global_bn): no statistics are stored
        return [site + "_bn/" + leaf + "_bn_scale", site + "_bn/" + leaf + "_bn_offset", None, None]

    def _sites(self):
        """(conv site, filter shape, has BN) in blob order"""
        if self.stem:
            yield self.STEM_SITE, (1, self.stem["k"], 1, self.stem["out"]), True
        for b in self.blocks:
            n, ci, co, k = b["name"], b["in"], b["out"], b["k"]
            yield n + "/branch1/conv1", (1, 1, ci, co), bool(b["i_bn"])
            yield n + "/branch2/conv2a", (1, 1, ci, co), True
            yield n + "/branch2/conv2b", (1, k, co, co), True
            yield n + "/branch2/conv2c", (1, 1, co, co), True

    def _rnn_and_head(self):
        H = self.hidden
        for l in range(self.rnn_layers):
            for d in ("fw", "bw"):
                yield self.lstm_scope(l, d) + "kernel", (self.lstm_in_width(l) + H, 4 * H)
                yield self.lstm_scope(l, d) + "bias", (4 * H,)
        yield "rnn_fnn_layer/weights", (2, H)
        yield "rnn_fnn_layer/bias", (H,)
        yield "rnn_fnn_layer/weights_class", (H, self.classes)
        yield "rnn_fnn_layer/bias_class", (self.classes,)

    def blob_layout(self):
        """Ordered {canonical name: shape} = the weight blob of include/chiron_amd.h: every BN site has the four slots
        scale, offset, pop_mean, pop_var whatever the BN mode (batch mode ignores the last two)."""
        v = OrderedDict()
        for site, shape, has_bn in self._sites():
            v[site + "/weights"] = shape
            if has_bn:
                for leaf in self.BN_LEAVES:
                    v[site + "_bn/" + leaf] = (shape[-1],)
        v.update(self._rnn_and_head())
        return v

    def variables(self):
        """Ordered {tf_variable_name: shape} a checkpoint of this model holds (what load_model reads).  Population BN:
        identical to blob_layout().  Batch BN (HEAD code): <site>_bn/<leaf>_bn_scale|_bn_offset, no statistics."""
        v = OrderedDict()
        for site, shape, has_bn in self._sites():
            v[site + "/weights"] = shape
            if has_bn:
                for name in self.bn_names(site):
                    if name is not None:
                        v[name] = (shape[-1],)
        v.update(self._rnn_and_head())
        return v

    def canonical_weights(self, weights):
        """{checkpoint or canonical names -> arrays}  ==>  OrderedDict under the canonical names of blob_layout().
        A HEAD-style (batch BN) checkpoint has no statistics: those slots are filled with mean 0 / variance 1."""
        out = OrderedDict()
        for site, shape, has_bn in self._sites():
            out[site + "/weights"] = weights[site + "/weights"]
            if not has_bn:
                continue
            for leaf, name in zip(self.BN_LEAVES, self.bn_names(site)):
                canon = site + "_bn/" + leaf
                if canon in weights:
                    out[canon] = weights[canon]
                elif name is not None and name in weights:
                    out[canon] = weights[name]
                elif name is None:
                    out[canon] = np.full(shape[-1], 1.0 if leaf == "pop_var" else 0.0, dtype=np.float32)
                else:
                    raise KeyError("weight %r (or %r) missing" % (name, canon))
        for name, _ in self._rnn_and_head():
            if name not in weights:
                raise KeyError("weight %r missing" % name)
            out[name] = weights[name]
        return out

    def pack(self, weights):
        """dict name -> array  ==>  flat float32 blob in ABI order."""
        canon = self.canonical_weights(weights)
        parts = []
        for name, shape in self.blob_layout().items():
            a = np.asarray(canon[name], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError("weight %r has shape %s, expected %s" % (name, a.shape, shape))
            parts.append(a.ravel())
        return np.ascontiguousarray(np.concatenate(parts))

    def output_len(self, segment_len):
        t = segment_len
        if self.stem:
            t = int(math.ceil(t / self.stem["stride"]))
        for b in self.blocks:
            t = int(math.ceil(t / b["stride"]))
        return t

    # -- C descriptor --------------------------------------------------------
    def to_c(self):
        d = _lib.ModelDesc()
        if len(self.blocks) > _lib.MAX_BLOCKS:
            raise ValueError("too many residual blocks")
        d.n_blocks = len(self.blocks)
        for i, b in enumerate(self.blocks):
            d.blocks[i] = _lib.ResBlock(b["in"], b["out"], b["k"], b["stride"], int(bool(b["i_bn"])))
        d.rnn_kind = _lib.RNN_STACK if self.rnn_kind == "stack" else _lib.RNN_MULTI
        d.rnn_layers = self.rnn_layers
        d.hidden = self.hidden
        d.classes = self.classes
        d.bn_mode = _lib.BN_POPULATION if self.bn_mode == "population" else _lib.BN_BATCH
        if self.stem:
            d.stem_k, d.stem_stride, d.stem_channels = self.stem["k"], self.stem["stride"], self.stem["out"]
        return d

    def to_dict(self):
        """Plain-dict form (what oracle/nn_oracle.py consumes in the tests)."""
        d = {"cnn": [dict(b) for b in self.blocks],
             "rnn": {"kind": self.rnn_kind, "layers": self.rnn_layers, "hidden": self.hidden},
             "bn_mode": self.bn_mode, "classes": self.classes}
        if self.stem:
            d["stem"] = dict(self.stem)
        return d


def dna_default_spec(bn_mode="population"):
    """cnn.py:380-389 DNA_model1 + rnn.py:20-97 (== shipped DNA_default graph)."""
    blocks = [{"name": "res_layer1", "in": 1, "out": 256, "k": 3, "stride": 1, "i_bn": True},
              {"name": "res_layer2", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False},
              {"name": "res_layer3", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False}]
    return ModelSpec(blocks, "stack", 3, 100, 5, bn_mode)


def rna_default_spec(bn_mode="population"):
    """The SHIPPED RNA_default graph (final.ckpt-80000.meta): three residual
    blocks, the first with a k=13 / stride-5 conv2b and stride-5 branch1, and
    MultiRNNCell stacks (rnn.py:99-174).  (HEAD's rna_model3, cnn.py:466-476,
    has no shipped weights -- SURVEY.md section 0 fact 4.)"""
    blocks = [{"name": "res_layer1", "in": 1, "out": 256, "k": 13, "stride": 5, "i_bn": True},
              {"name": "res_layer2", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False},
              {"name": "res_layer3", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False}]
    return ModelSpec(blocks, "multi", 3, 100, 5, bn_mode)


def rna_head_spec(model="rna_model3", bn_mode="population"):
    """HEAD's RNA_model2 / RNA_model3 (cnn.py:454-476): a strided stem convolution on the signal
    (k, stride = 9, 5 / 14, 7) + BN + ReLU, then three stride-1 residual blocks of 256 channels (res_layer1 with BN on
    its shortcut), MultiRNNCell stacks.  No shipped checkpoint uses it (the shipped RNA_default weights belong to
    rna_default_spec()); it is what HEAD builds for a newly trained RNA model."""
    k, stride = {"rna_model2": (9, 5), "rna_model3": (14, 7)}[model]
    blocks = [{"name": "res_layer1", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": True},
              {"name": "res_layer2", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False},
              {"name": "res_layer3", "in": 256, "out": 256, "k": 3, "stride": 1, "i_bn": False}]
    return ModelSpec(blocks, "multi", 3, 100, 5, bn_mode, stem={"k": k, "stride": stride, "out": 256})


def read_config(config_file):
    """chiron_model.py:37-48."""
    if config_file is not None:
        with open(config_file) as f:
            return json.load(f)
    return {"cnn": {"model": "dna_model1"},
            "rnn": {"layer_num": 3, "hidden_num": 100, "cell_type": "LSTM", "layer_type": "normal"},
            "opt_method": "Adam", "fl_gamma": 2}


def spec_from_config(config, bn_mode="population"):
    """Topology for a model.json when no checkpoint index is available."""
    rnn = config["rnn"]
    if rnn.get("cell_type", "LSTM") != "LSTM":
        raise ValueError("Cell type unrecognized.")          # rnn.py:58
    name = config["cnn"]["model"]
    if name == "dna_model1":
        spec = dna_default_spec(bn_mode)
    elif name in ("rna_model3", "rna_shipped"):
        # RNA_default/model.json says rna_model3, but its checkpoint holds the k=13 / stride-5 graph (SURVEY.md section 0
        # fact 4); with a checkpoint index present spec_from_variables decides, this branch is the index-less default
        spec = rna_default_spec(bn_mode)
    elif name in ("rna_model2", "rna_model3_head"):
        spec = rna_head_spec("rna_model2" if name == "rna_model2" else "rna_model3", bn_mode)
    else:
        raise ValueError("CNN model %r has no shipped weights and is not supported" % name)
    spec.rnn_layers = int(rnn["layer_num"])
    spec.hidden = int(rnn["hidden_num"])
    spec.rnn_kind = "multi" if rnn.get("layer_type") == "rna" else "stack"   # chiron_model.py:157
    return spec


def spec_from_variables(shapes, strides=None):
    """Derive the topology from checkpoint variable names -> shapes.

    strides: optional {block_name: stride}; when absent, a k=13 conv2b means
    the shipped RNA block (stride 5, from the .meta Conv2D attrs), else 1."""
    names = set(shapes)
    stem = None
    if (ModelSpec.STEM_SITE + "/weights") in names:
        ws = shapes[ModelSpec.STEM_SITE + "/weights"]
        k = int(ws[1])
        # the stride is a graph attribute, not a variable: HEAD pairs k = 9 with 5 and k = 14 with 7 (cnn.py:456, :468)
        stem = {"k": k, "stride": int((strides or {}).get("conv_layer", {9: 5, 14: 7}.get(k, 1))), "out": int(ws[3])}
    blocks = []
    i = 1
    while ("res_layer%d/branch2/conv2b/weights" % i) in names:
        n = "res_layer%d" % i
        w2b = shapes[n + "/branch2/conv2b/weights"]
        w1 = shapes[n + "/branch1/conv1/weights"]
        k = int(w2b[1])
        stride = (strides or {}).get(n, 5 if k == 13 else 1)
        i_bn = (n + "/branch1/conv1_bn/scale") in names or (n + "/branch1/conv1_bn/conv1_bn_scale") in names
        blocks.append({"name": n, "in": int(w1[2]), "out": int(w1[3]), "k": k, "stride": int(stride), "i_bn": i_bn})
        i += 1
    if not blocks:
        raise ValueError("no res_layerN variables found")
    bn_mode = "population" if ("res_layer1/branch2/conv2a_bn/pop_mean" in names) else "batch"
    if any(x.startswith("BDGRU_rnn/") for x in names):
        kind, pat = "multi", "BDGRU_rnn/fw/multi_rnn_cell/cell_%d/lstm_cell/bias"
    else:
        kind, pat = "stack", "BDLSTM_rnn/cell_%d/bidirectional_rnn/fw/lstm_cell/bias"
    layers = 0
    while (pat % layers) in names:
        layers += 1
    hidden = int(shapes[pat % 0][0]) // 4
    classes = int(shapes["rnn_fnn_layer/bias_class"][0])
    return ModelSpec(blocks, kind, layers, hidden, classes, bn_mode, stem=stem)


# ---------------------------------------------------------------------------
# Seeded synthetic weights (the trained *.data files are stripped from the
# reference tree: .MISSING_LARGE_BLOBS).  Exact checkpoint shapes; BN
# statistics chosen so activations stay O(1) for raw DAC-count input
# (SURVEY.md 8d "Synthetic inputs").
# ---------------------------------------------------------------------------
SIGNAL_MEAN = 500.0
SIGNAL_STD = 80.0


def synthetic_weights(spec, seed=1234, logit_gain=20.0, lstm_gain=3.0):
    rng = np.random.RandomState(seed)
    w = OrderedDict()
    H = spec.hidden

    def bn(site, co, mean, var):
        w[site + "_bn/scale"] = rng.uniform(0.9, 1.1, co).astype(np.float32)
        w[site + "_bn/offset"] = rng.normal(0, 0.1, co).astype(np.float32)
        w[site + "_bn/pop_mean"] = np.asarray(mean, dtype=np.float32)
        w[site + "_bn/pop_var"] = np.asarray(var, dtype=np.float32)

    # Second-moment tracking keeps every BN site's pop_var close to the variance its input really
    # has, so activations neither blow up nor saturate the LSTM gates (constants calibrated once
    # against the float64 oracle on synthetic_signal()).
    m2 = 1.0  # E[x^2] of the block input
    if spec.stem:
        k, co = spec.stem["k"], spec.stem["out"]
        ws = rng.normal(0, math.sqrt(2.0 / (k + co)), (1, k, 1, co)).astype(np.float32)
        w[spec.STEM_SITE + "/weights"] = ws
        tap_sum = ws.reshape(k, co).sum(axis=0)
        # the squiggle is piecewise constant over ~9 samples: neighbouring taps see almost the same level
        bn(spec.STEM_SITE, co, SIGNAL_MEAN * tap_sum * rng.uniform(0.97, 1.03, co),
           (SIGNAL_STD * 0.75) ** 2 * (tap_sum ** 2 + (ws.reshape(k, co) ** 2).sum(axis=0)) * rng.uniform(0.8, 1.2, co) + 1e-3)
        m2 = 0.55
    for b in spec.blocks:
        n, ci, co, k = b["name"], b["in"], b["out"], b["k"]
        xav = lambda fan_in, fan_out, shape: rng.normal(0, math.sqrt(2.0 / (fan_in + fan_out)), shape).astype(np.float32)
        if ci == 1:
            w1 = xav(1, co, (1, 1, 1, co))
            w[n + "/branch1/conv1/weights"] = w1
            if b["i_bn"]:
                bn(n + "/branch1/conv1", co, SIGNAL_MEAN * w1.ravel() * rng.uniform(0.97, 1.03, co),
                   (SIGNAL_STD * w1.ravel()) ** 2 * rng.uniform(0.8, 1.2, co) + 1e-3)
            w2a = xav(1, co, (1, 1, 1, co))
            w[n + "/branch2/conv2a/weights"] = w2a
            bn(n + "/branch2/conv2a", co, SIGNAL_MEAN * w2a.ravel() * rng.uniform(0.97, 1.03, co),
               (SIGNAL_STD * w2a.ravel()) ** 2 * rng.uniform(0.8, 1.2, co) + 1e-3)
            var_b1 = 1.0
        else:
            g1 = 0.5   # un-normalised shortcut (no BN on branch1 after res_layer1): damp it
            w[n + "/branch1/conv1/weights"] = g1 * xav(ci, co, (1, 1, ci, co))
            var_b1 = g1 * g1 * m2 * 2.0 * ci / (ci + co)
            if b["i_bn"]:
                bn(n + "/branch1/conv1", co, rng.normal(0, 0.1, co), var_b1 * rng.uniform(0.8, 1.2, co))
                var_b1 = 1.0
            w[n + "/branch2/conv2a/weights"] = xav(ci, co, (1, 1, ci, co))
            bn(n + "/branch2/conv2a", co, rng.normal(0.0, 0.1, co), 0.65 * m2 * 2.0 * ci / (ci + co) * rng.uniform(0.8, 1.2, co))
        m2a = 0.55   # E[x^2] after BN(~N(0,1)) + ReLU
        w[n + "/branch2/conv2b/weights"] = xav(k * co, co, (1, k, co, co))
        bn(n + "/branch2/conv2b", co, rng.normal(0.0, 0.1, co), m2a * 2.0 * k / (k + 1) * rng.uniform(0.8, 1.2, co))
        w[n + "/branch2/conv2c/weights"] = xav(co, co, (1, 1, co, co))
        bn(n + "/branch2/conv2c", co, rng.normal(0.0, 0.1, co), m2a * rng.uniform(0.8, 1.2, co))
        m2 = 0.5 * (1.0 + var_b1) + 0.1
    for l in range(spec.rnn_layers):
        for d in ("fw", "bw"):
            rows = spec.lstm_in_width(l) + H
            lim = lstm_gain / math.sqrt(rows)
            w[spec.lstm_scope(l, d) + "kernel"] = rng.uniform(-lim, lim, (rows, 4 * H)).astype(np.float32)
            w[spec.lstm_scope(l, d) + "bias"] = rng.uniform(-0.1, 0.1, 4 * H).astype(np.float32)
    w["rnn_fnn_layer/weights"] = rng.normal(0, math.sqrt(2.0 / (2 * H)), (2, H)).astype(np.float32)
    w["rnn_fnn_layer/bias"] = rng.normal(0, 0.05, H).astype(np.float32)
    w["rnn_fnn_layer/weights_class"] = (logit_gain * rng.normal(0, math.sqrt(2.0 / H), (H, spec.classes))).astype(np.float32)
    w["rnn_fnn_layer/bias_class"] = rng.normal(0, 0.1, spec.classes).astype(np.float32)
    # keep the variable order / set identical to the spec's contract
    ordered = OrderedDict((k, w[k]) for k in spec.blob_layout())
    return ordered


def synthetic_signal(n_reads, n_samples, seed=1234):
    """Piecewise-constant 'squiggle' in raw DAC counts: dwell ~
    Geometric(mean 8.9 samples), level ~ N(500,80^2) clipped to [200,1000],
    + N(0,8^2) noise, rounded to integers like int16 fast5 signal."""
    out = np.empty((n_reads, n_samples), dtype=np.float32)
    for r in range(n_reads):
        rng = np.random.RandomState(seed + r)
        n_ev = int(n_samples / 8.9 * 1.5) + 16
        dwell = rng.geometric(1.0 / 8.9, n_ev)
        level = np.clip(rng.normal(SIGNAL_MEAN, SIGNAL_STD, n_ev), 200, 1000)
        sig = np.repeat(level, dwell)[:n_samples]
        while sig.shape[0] < n_samples:       # pathological tail: extend
            sig = np.concatenate([sig, sig])[:n_samples]
        out[r] = np.rint(sig + rng.normal(0, 8.0, n_samples))
    return out


def load_model(model_dir, allow_synthetic=False, seed=1234):
    """(spec, weights) for a reference-style model folder (model.json +
    checkpoint + .index/.data bundle).  Replaces Saver.restore
    (chiron_eval.py:272-276).  The reference tree ships no .data files; with
    allow_synthetic the exact-shape seeded weights are used instead."""
    from . import tf_bundle
    config = read_config(os.path.join(model_dir, "model.json"))
    prefix = tf_bundle.latest_checkpoint(model_dir)
    if prefix is not None and os.path.exists(prefix + ".index"):
        entries = tf_bundle.read_index(prefix + ".index")
        shapes = {k: v["shape"] for k, v in entries.items() if k}
        spec = spec_from_variables(shapes)
        if os.path.exists(prefix + ".data-00000-of-00001"):
            weights = spec.canonical_weights(tf_bundle.read_tensors(prefix, entries, list(spec.variables())))
            return spec, weights, config
    else:
        spec = spec_from_config(config)
    if not allow_synthetic:
        raise FileNotFoundError("no checkpoint data under %s (the reference tree strips *.data-00000-of-00001); "
                                "pass allow_synthetic=True / --synthetic-weights to run with seeded weights" % model_dir)
    return spec, synthetic_weights(spec, seed), config
