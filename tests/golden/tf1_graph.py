"""TEST INFRASTRUCTURE (fixture generation only; runs in the build container, where /root/reference exists).

A reader and a small numpy evaluator for TensorFlow-1.x `MetaGraphDef` files, used by
`make_meta_golden.py` to turn the graphs the reference ships
(`chiron/model/DNA_default/final.ckpt-158301.meta`, `chiron/model/RNA_default/final.ckpt-80000.meta`)
into fixtures.  The evaluator executes the REFERENCE'S OWN NODE LIST -- `tf.cond` as Switch/Merge with dead-branch
propagation, `tf.while_loop` (dynamic_rnn) as Enter/Merge/Switch/NextIteration/Exit frames with TensorArrays --
so what it computes is the composition recorded by the reference (which conv feeds which BN, gate order, forget
bias, sequence-length masking, ReverseSequence placement, the FC head), not a restatement of it.  Only the
per-op arithmetic (Conv2D SAME padding, Sigmoid, ...) is this file's; every op is a few numpy lines below.

Nothing here is imported by the product or by the GPU-side tests; the generated fixtures are.
"""
import struct

import numpy as np


# ------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, p):
    v = 0
    shift = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, p
        shift += 7


def _fields(buf):
    p = 0
    n = len(buf)
    while p < n:
        tag, p = _varint(buf, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        elif wt == 2:
            ln, p = _varint(buf, p)
            v = buf[p:p + ln]
            p += ln
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _sint(v):
    return v if v < (1 << 63) else v - (1 << 64)


def _packed_ints(wt, v):
    if wt == 0:
        return [_sint(v)]
    out = []
    p = 0
    while p < len(v):
        x, p = _varint(v, p)
        out.append(_sint(x))
    return out


_NP = {1: "<f4", 2: "<f8", 3: "<i4", 9: "<i8", 10: "?"}   # tensorflow DataType enum
DT_NAMES = {1: "float32", 2: "float64", 3: "int32", 7: "string", 9: "int64", 10: "bool", 20: "resource"}


def _shape(buf):
    dims = []
    for f, _, v in _fields(buf):
        if f == 2:
            size = 0
            for f3, _, v3 in _fields(v):
                if f3 == 1:
                    size = _sint(v3)
            dims.append(size)
    return dims


def _tensor(buf):
    dtype, shape, content, vals = 0, [], None, []
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = _shape(v)
        elif f == 4:
            content = bytes(v)
        elif f == 5:      # float_val
            vals += [struct.unpack("<f", v)[0]] if wt == 5 else list(np.frombuffer(bytes(v), "<f4"))
        elif f == 6:      # double_val
            vals += [struct.unpack("<d", v)[0]] if wt == 1 else list(np.frombuffer(bytes(v), "<f8"))
        elif f in (7, 10, 11):   # int_val / int64_val / bool_val
            vals += _packed_ints(wt, v)
        elif f == 8:
            vals.append(bytes(v))
    if dtype == 7:
        return vals
    npd = np.dtype(_NP[dtype])
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        return np.frombuffer(content, npd).reshape(shape).copy()
    if not vals:
        return np.zeros(shape, npd)
    if len(vals) < n:     # TensorProto convention: the last value repeats
        vals = vals + [vals[-1]] * (n - len(vals))
    return np.array(vals, npd).reshape(shape)


def _attr(buf):
    for f, wt, v in _fields(buf):
        if f == 1:
            out = {}
            for f2, wt2, v2 in _fields(v):
                if f2 == 2:
                    out.setdefault("s", []).append(bytes(v2).decode("latin1"))
                elif f2 == 3:
                    out.setdefault("i", []).extend(_packed_ints(wt2, v2))
                elif f2 == 4:
                    out.setdefault("f", []).extend(
                        [struct.unpack("<f", v2)[0]] if wt2 == 5 else [float(x) for x in np.frombuffer(bytes(v2), "<f4")])
                elif f2 == 5:
                    out.setdefault("b", []).extend([bool(x) for x in _packed_ints(wt2, v2)])
                elif f2 == 6:
                    out.setdefault("type", []).extend(_packed_ints(wt2, v2))
                elif f2 == 7:
                    out.setdefault("shape", []).append(_shape(v2))
            for key in ("i", "s", "f", "b", "type", "shape"):
                if key in out:
                    return out[key]
            return []
        if f == 2:
            return bytes(v).decode("latin1")
        if f == 3:
            return _sint(v)
        if f == 4:
            return struct.unpack("<f", v)[0]
        if f == 5:
            return bool(v)
        if f == 6:
            return ("dtype", v)
        if f == 7:
            return ("shape", _shape(v))
        if f == 8:
            return _tensor(v)
    return None


class Node(object):
    __slots__ = ("name", "op", "inputs", "control", "attr")

    def __init__(self):
        self.name = self.op = None
        self.inputs = []     # [(node name, output index)]
        self.control = []    # [node name]
        self.attr = {}


def load_meta_graph(path):
    """MetaGraphDef{2: graph_def{1: node*}} -> {name: Node} in file order."""
    buf = memoryview(open(path, "rb").read())
    nodes = {}
    for f, _, v in _fields(buf):
        if f != 2:
            continue
        for f2, _, v2 in _fields(v):
            if f2 != 1:
                continue
            n = Node()
            for f3, _, v3 in _fields(v2):
                if f3 == 1:
                    n.name = bytes(v3).decode()
                elif f3 == 2:
                    n.op = bytes(v3).decode()
                elif f3 == 3:
                    s = bytes(v3).decode()
                    if s.startswith("^"):
                        n.control.append(s[1:])
                    else:
                        name, _, idx = s.partition(":")
                        n.inputs.append((name, int(idx) if idx else 0))
                elif f3 == 5:
                    key = val = None
                    for f4, _, v4 in _fields(v3):
                        if f4 == 1:
                            key = bytes(v4).decode()
                        elif f4 == 2:
                            val = _attr(v4)
                    n.attr[key] = val
            nodes[n.name] = n
    return nodes


# ------------------------------------------------------------------------------------------------- evaluator
class _Dead(object):
    def __repr__(self):
        return "DEAD"


DEAD = _Dead()


class _TensorArray(object):
    def __init__(self, size, element_shape, dtype):
        self.items = [None] * size
        self.element_shape = element_shape
        self.dtype = dtype

    def read(self, i):
        v = self.items[i]
        if v is None:
            # TF: reading a never-written element of a TensorArray whose element_shape is fully defined gives zeros
            if self.element_shape is None or any(d < 0 for d in self.element_shape):
                raise ValueError("TensorArray element %d was never written and its shape is unknown" % i)
            return np.zeros(self.element_shape, self.dtype)
        return v


def same_pad(width, k, stride):
    out = -(-width // stride)
    total = max((out - 1) * stride + k - width, 0)
    return out, total // 2, total - total // 2


def conv2d_nhwc(x, w, strides, padding):
    """tf.nn.conv2d, NHWC x [N,H,W,C], HWIO w, cross-correlation; only H == 1 filters of height 1 occur here."""
    n, h, wid, c = x.shape
    kh, kw, cin, cout = w.shape
    if kh != 1 or h != 1 or strides[1] != 1 or strides[0] != 1 or strides[3] != 1:
        raise NotImplementedError("only 1 x k convolutions over H == 1")
    s = strides[2]
    if padding == "SAME":
        out, left, right = same_pad(wid, kw, s)
    elif padding == "VALID":
        out, left, right = (wid - kw) // s + 1, 0, 0
    else:
        raise ValueError(padding)
    xp = np.zeros((n, wid + left + right, c), x.dtype)
    xp[:, left:left + wid] = x[:, 0]
    y = np.zeros((n, out, cout), x.dtype)
    for tap in range(kw):
        y += xp[:, tap:tap + (out - 1) * s + 1:s] @ w[0, tap]
    return y[:, None]


def strided_slice(x, begin, end, strides, begin_mask, end_mask, shrink_mask):
    idx = []
    for d in range(len(begin)):
        if shrink_mask >> d & 1:
            idx.append(int(begin[d]))
            continue
        b = None if begin_mask >> d & 1 else int(begin[d])
        e = None if end_mask >> d & 1 else int(end[d])
        idx.append(slice(b, e, int(strides[d])))
    return np.asarray(x)[tuple(idx)]


def reverse_sequence(x, lens, seq_dim, batch_dim):
    if batch_dim != 0 or seq_dim != 1:
        raise NotImplementedError
    y = x.copy()
    for b in range(x.shape[0]):
        n = int(lens[b])
        y[b, :n] = x[b, :n][::-1]
    return y


class GraphEval(object):
    """Evaluate tensors of a TF-1 graph with numpy.  `feeds`: {node name: value} (placeholders AND variables);
    float tensors are computed in `float_dtype` (float64 for fixtures).  Assign-type ops are never executed."""

    def __init__(self, nodes, feeds, float_dtype=np.float64):
        self.nodes = nodes
        self.feeds = feeds
        self.fd = np.dtype(float_dtype)
        self.memo = {}
        self.frame_of = self._assign_frames()
        self.frame_results = {}
        self.op_counts = {}

    # -- which while-frame a node lives in (None = top level): forward propagation from the Enter nodes
    def _assign_frames(self):
        frame = {}
        consumers = {}
        for n in self.nodes.values():
            for src, _ in n.inputs:
                consumers.setdefault(src, []).append(n.name)
            for src in n.control:
                consumers.setdefault(src, []).append(n.name)
        work = []
        for n in self.nodes.values():
            if n.op == "Enter":
                frame[n.name] = n.attr["frame_name"]
                work.append(n.name)
        while work:
            cur = work.pop()
            if self.nodes[cur].op == "Exit":
                continue
            for c in consumers.get(cur, ()):
                if c in frame or self.nodes[c].op == "Enter":
                    continue
                frame[c] = frame[cur]
                work.append(c)
        for n in self.nodes.values():        # an Exit leaves its frame
            if n.op == "Exit":
                frame.pop(n.name, None)
        return frame

    def _cast(self, a):
        a = np.asarray(a)
        return a.astype(self.fd) if a.dtype.kind == "f" else a

    def run(self, name, index=0):
        return self._ev(name, None)[index]

    # -- evaluation of one node in a context (ctx = per-iteration memo of the active frame, or None)
    def _ev(self, name, ctx):
        node = self.nodes[name]
        fr = None if node.op == "Enter" else self.frame_of.get(name)
        memo = self.memo if fr is None else ctx
        if fr is not None and ctx is None:
            raise RuntimeError("%s lives in frame %s but was requested from outside" % (name, fr))
        if name in memo:
            return memo[name]
        out = self._compute(node, ctx if fr is not None else None)
        memo[name] = out
        return out

    def _in(self, node, ctx, k):
        src, idx = node.inputs[k]
        return self._ev(src, ctx)[idx]

    def _compute(self, node, ctx):
        op = node.op
        self.op_counts[op] = self.op_counts.get(op, 0) + 1
        if node.name in self.feeds:
            return (self._cast(self.feeds[node.name]),)
        if op == "Exit":
            return (self._run_frame(node),)
        if op == "Enter":        # evaluated in the enclosing context, constant for the frame's lifetime
            src, idx = node.inputs[0]
            return (self._ev(src, None)[idx],)
        if op == "Merge":
            if self.frame_of.get(node.name) is not None:
                raise RuntimeError("loop Merge %s evaluated outside _run_frame" % node.name)
            for k in range(len(node.inputs)):
                v = self._in(node, ctx, k)
                if v is not DEAD:
                    return (v, k)
            return (DEAD, -1)
        if op in ("Placeholder", "VariableV2", "VarHandleOp"):
            raise KeyError("no value fed for %s (%s)" % (node.name, op))
        if op in ("Assign", "AssignSub", "AssignAdd", "NoOp", "Assert"):
            # side effects are never executed; deadness still propagates so control-dependent constants die with their branch
            args = [self._in(node, ctx, k) for k in range(len(node.inputs))]
            return (DEAD if any(a is DEAD for a in args) else None,)
        args = [self._in(node, ctx, k) for k in range(len(node.inputs))]
        ctl_dead = False
        for c in node.control:
            if self.nodes[c].op in ("Assert",):
                continue
            if self._ev(c, ctx)[0] is DEAD:
                ctl_dead = True
        if op in ("Switch", "RefSwitch"):
            data, pred = args
            if data is DEAD or pred is DEAD:
                return (DEAD, DEAD)
            return (DEAD, data) if bool(pred) else (data, DEAD)
        if ctl_dead or any(a is DEAD for a in args):
            n_out = 4 if op == "Split" else 2
            return (DEAD,) * n_out
        fn = getattr(self, "_op_" + op, None)
        if fn is None:
            raise NotImplementedError("op %s (%s)" % (op, node.name))
        out = fn(node, *args)
        return out if isinstance(out, tuple) else (out,)

    # -- tf.while_loop: iterate the frame's body until LoopCond is false; returns the value of the asked Exit
    def _run_frame(self, exit_node):
        switch = self.nodes[exit_node.inputs[0][0]]
        merge_name = switch.inputs[0][0]
        fname = self.frame_of[merge_name]
        if fname not in self.frame_results:
            merges = [n for n in self.nodes.values() if n.op == "Merge" and self.frame_of.get(n.name) == fname]
            loopcond = [n for n in self.nodes.values() if n.op == "LoopCond" and self.frame_of.get(n.name) == fname]
            assert len(loopcond) == 1
            state = {}
            nexts = {}
            for m in merges:
                (a, ai), (b, bi) = m.inputs
                enter, nxt = (a, b) if self.nodes[a].op == "Enter" else (b, a)
                assert self.nodes[enter].op == "Enter" and self.nodes[nxt].op == "NextIteration"
                state[m.name] = self._ev(enter, None)[0]
                nexts[m.name] = self.nodes[nxt].inputs[0]
            iters = 0
            while True:
                ctx = {m: (v, 0) for m, v in state.items()}
                pred = self._ev(loopcond[0].name, ctx)[0]
                if not bool(pred):
                    break
                new_state = {}
                for m, (src, idx) in nexts.items():
                    new_state[m] = self._ev(src, ctx)[idx]
                state = new_state
                iters += 1
            self.frame_results[fname] = (state, iters)
        return self.frame_results[fname][0][merge_name]

    # -- ops ------------------------------------------------------------------------------------------------
    def _op_Const(self, n):
        v = n.attr["value"]
        return self._cast(v) if isinstance(v, np.ndarray) else v

    def _op_Identity(self, n, x):
        return x

    _op_StopGradient = _op_Identity
    _op_LoopCond = _op_Identity
    _op_NextIteration = _op_Identity

    def _op_Add(self, n, a, b):
        return a + b

    def _op_Sub(self, n, a, b):
        return a - b

    def _op_Mul(self, n, a, b):
        return a * b

    def _op_Maximum(self, n, a, b):
        return np.maximum(a, b)

    def _op_Minimum(self, n, a, b):
        return np.minimum(a, b)

    def _op_Less(self, n, a, b):
        return np.less(a, b)

    def _op_GreaterEqual(self, n, a, b):
        return np.greater_equal(a, b)

    def _op_LogicalAnd(self, n, a, b):
        return np.logical_and(a, b)

    def _op_SquaredDifference(self, n, a, b):
        return (a - b) * (a - b)

    def _op_Rsqrt(self, n, x):
        return 1.0 / np.sqrt(x)

    def _op_Relu(self, n, x):
        return np.maximum(x, 0)

    def _op_Sigmoid(self, n, x):
        return 1.0 / (1.0 + np.exp(-x))

    def _op_Tanh(self, n, x):
        return np.tanh(x)

    def _op_Select(self, n, c, t, e):
        c = np.asarray(c)
        if c.ndim == 1 and np.ndim(t) > 1:      # tf.where with a vector condition selects whole rows
            c = c.reshape((-1,) + (1,) * (np.ndim(t) - 1))
        return np.where(c, t, e)

    def _op_Mean(self, n, x, axes):
        return np.mean(x, axis=tuple(int(a) for a in np.atleast_1d(axes)), keepdims=bool(n.attr.get("keep_dims")))

    def _op_Sum(self, n, x, axes):
        return np.sum(x, axis=tuple(int(a) for a in np.atleast_1d(axes)), keepdims=bool(n.attr.get("keep_dims")))

    def _op_Max(self, n, x, axes):
        return np.max(x, axis=tuple(int(a) for a in np.atleast_1d(axes)), keepdims=bool(n.attr.get("keep_dims")))

    def _op_Squeeze(self, n, x):
        dims = n.attr.get("squeeze_dims") or None
        return np.squeeze(x, axis=tuple(dims) if dims else None)

    def _op_Reshape(self, n, x, shape):
        return np.reshape(x, [int(s) for s in shape])

    def _op_Transpose(self, n, x, perm):
        return np.transpose(x, [int(p) for p in perm])

    def _op_Pack(self, n, *xs):
        return np.stack([np.asarray(x) for x in xs], axis=n.attr.get("axis", 0))

    def _op_ConcatV2(self, n, *xs):
        return np.concatenate([np.atleast_1d(x) for x in xs[:-1]], axis=int(xs[-1]))

    def _op_Range(self, n, start, limit, delta):
        return np.arange(int(start), int(limit), int(delta), dtype=np.int32)

    def _op_Fill(self, n, dims, value):
        return np.full([int(d) for d in dims], value)

    def _op_Shape(self, n, x):
        return np.array(np.shape(x), np.int32)

    def _op_StridedSlice(self, n, x, begin, end, strides):
        if n.attr.get("ellipsis_mask") or n.attr.get("new_axis_mask"):
            raise NotImplementedError
        return strided_slice(x, begin, end, strides, n.attr.get("begin_mask", 0), n.attr.get("end_mask", 0),
                             n.attr.get("shrink_axis_mask", 0))

    def _op_Split(self, n, dim, x):
        return tuple(np.split(x, n.attr["num_split"], axis=int(dim)))

    def _op_MatMul(self, n, a, b):
        if n.attr.get("transpose_a"):
            a = a.T
        if n.attr.get("transpose_b"):
            b = b.T
        return a @ b

    def _op_BiasAdd(self, n, x, b):
        return x + b

    def _op_Conv2D(self, n, x, w):
        if n.attr.get("data_format", "NHWC") != "NHWC" or any(d != 1 for d in n.attr.get("dilations", [1])):
            raise NotImplementedError
        return conv2d_nhwc(x, w, n.attr["strides"], n.attr["padding"])

    def _op_ReverseSequence(self, n, x, lens):
        return reverse_sequence(x, lens, n.attr["seq_dim"], n.attr.get("batch_dim", 0))

    def _op_TensorArrayV3(self, n, size):
        es = n.attr.get("element_shape")
        es = es[1] if isinstance(es, tuple) else None
        dt = n.attr.get("dtype")
        npd = self.fd if (isinstance(dt, tuple) and dt[1] in (1, 2)) else np.dtype(_NP.get(dt[1], "<f4")) if isinstance(dt, tuple) else self.fd
        return (_TensorArray(int(size), es, npd), np.zeros((), self.fd))

    def _op_TensorArrayScatterV3(self, n, handle, indices, value, flow):
        for k, i in enumerate(indices):
            handle.items[int(i)] = value[k]
        return flow

    def _op_TensorArrayReadV3(self, n, handle, index, flow):
        return handle.read(int(index))

    def _op_TensorArrayWriteV3(self, n, handle, index, value, flow):
        handle.items[int(index)] = value
        return flow

    def _op_TensorArrayGatherV3(self, n, handle, indices, flow):
        return np.stack([handle.read(int(i)) for i in indices])

    def _op_TensorArraySizeV3(self, n, handle, flow):
        return np.int32(len(handle.items))


# ------------------------------------------------------------------------------- symbolic view (for the digest)
class Symbolic(object):
    """Readable record of a sub-graph: nested-call strings over `leaves` (tensor key -> symbol).  Identity / Enter are
    looked through, variables and placeholders print by name, small constants by value, the Switch of a while-loop
    variable prints as the variable itself, and any non-trivial tensor used more than once among the requested roots
    is bound to a name (t1, t2, ...) so shared sub-expressions -- the LSTM pre-activation feeding four gates -- are
    written once."""

    def __init__(self, nodes, leaves=None, through_cond=False):
        self.nodes = nodes
        self.leaves = dict(leaves or {})
        self.through_cond = through_cond     # print tf.cond Switches as their data input; ports seen go to cond_ports
        self.cond_ports = set()

    @staticmethod
    def _key(name, index):
        return name if index == 0 else "%s:%d" % (name, index)

    def _resolve(self, name, index):
        """look through Identity / Enter / loop Switch; -> (name, index)"""
        while True:
            if self._key(name, index) in self.leaves:
                return name, index
            n = self.nodes[name]
            if n.op in ("Identity", "Enter", "StopGradient", "LoopCond", "NextIteration"):
                name, index = n.inputs[0]
            elif n.op == "Switch" and self.nodes[n.inputs[0][0]].op == "Merge" and \
                    any(self.nodes[s].op == "NextIteration" for s, _ in self.nodes[n.inputs[0][0]].inputs):
                name, index = n.inputs[0]          # Switch(loop var, LoopCond): the loop variable itself
            elif n.op == "Switch" and self.through_cond:
                self.cond_ports.add(index)
                name, index = n.inputs[0]
            else:
                return name, index

    def _atom(self, name, index):
        key = self._key(name, index)
        if key in self.leaves:
            return self.leaves[key]
        n = self.nodes[name]
        if n.op == "Const":
            v = n.attr["value"]
            if isinstance(v, np.ndarray) and v.size <= 4:
                return repr(v.tolist())
            return "Const%s" % (list(v.shape),)
        if n.op == "VariableV2":
            return "var(%s)" % name
        if n.op == "Placeholder":
            return "placeholder(%s)" % name
        if n.op == "Merge" and any(self.nodes[s].op == "NextIteration" for s, _ in n.inputs):
            return "loopvar(%s)" % name
        return None

    def render(self, roots):
        """roots: {label: (name, index)} -> {"let": [[symbol, expr], ...], label: expr, ...}"""
        uses = {}
        order = []

        def visit(name, index):
            name, index = self._resolve(name, index)
            if self._atom(name, index) is not None:
                return
            uses[name] = uses.get(name, 0) + 1
            if uses[name] > 1:
                return
            for s, i in self.nodes[name].inputs:
                visit(s, i)
            order.append(name)

        for nm, ix in roots.values():
            visit(nm, ix)
        bound = {}
        lets = []

        def text(name, index, top=False):
            name, index = self._resolve(name, index)
            atom = self._atom(name, index)
            if atom is not None:
                return atom
            n = self.nodes[name]
            multi = n.op in ("Split", "Switch", "RefSwitch", "TensorArrayV3", "Merge")
            suffix = ":%d" % index if (multi or index) else ""
            if name in bound and not top:
                return bound[name] + suffix
            extra = ""
            if n.op == "Conv2D":
                extra = "{strides=%s,padding=%s}" % (n.attr["strides"], n.attr["padding"])
            elif n.op == "Split":
                extra = "{num_split=%d}" % n.attr["num_split"]
            elif n.op == "ReverseSequence":
                extra = "{seq_dim=%d,batch_dim=%d}" % (n.attr["seq_dim"], n.attr.get("batch_dim", 0))
            elif n.op in ("Mean", "Sum", "Max"):
                extra = "{keep_dims=%s}" % bool(n.attr.get("keep_dims"))
            elif n.op == "MatMul" and (n.attr.get("transpose_a") or n.attr.get("transpose_b")):
                extra = "{ta=%s,tb=%s}" % (bool(n.attr.get("transpose_a")), bool(n.attr.get("transpose_b")))
            body = "%s%s(%s)" % (n.op, extra, ", ".join(text(s, i) for s, i in n.inputs))
            return body if top else body + suffix

        for name in order:                      # topological: inputs first
            if uses[name] > 1:
                sym = "t%d" % (len(lets) + 1)
                lets.append([sym, text(name, 0, top=True)])
                bound[name] = sym
        out = {"let": lets}
        for label, (nm, ix) in roots.items():
            out[label] = text(nm, ix)
        return out


def expression(nodes, name, index=0, leaves=None, sym=None):
    """single-root convenience form of Symbolic.render: the expression string, with 'where' bindings appended"""
    r = (sym or Symbolic(nodes, leaves)).render({"e": (name, index)})
    if r["let"]:
        return r["e"] + " where " + "; ".join("%s = %s" % (a, b) for a, b in r["let"])
    return r["e"]
