#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container, where /root/reference exists.  It imports the
reference's pure-Python host functions (with tensorflow / h5py / statsmodels /
Bio replaced by MagicMock stubs, and the two numpy aliases the reference still
uses restored), feeds them small inputs, and records input -> output pairs as
data.  No reference source text is written anywhere; only vectors.

    python tests/golden/make_golden.py

Outputs:
    tests/golden/host_golden.json      function-level known-answer vectors
    tests/golden/example_dna/...       the reference's own checked-in example
                                       data files (raw signal of read1,
                                       segments + result of reads 1-5,
                                       read1.fast5) -- data fixtures.
"""
import json
import os
import random
import shutil
import sys
from collections import namedtuple
from unittest import mock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(REF, "chiron", "example_data", "DNA")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import importlib.abc
    import importlib.machinery

    class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        """Any import below these absent third-party roots resolves to a MagicMock."""
        roots = ("tensorflow", "h5py", "statsmodels", "Bio", "mappy")

        def find_spec(self, fullname, path, target=None):
            if fullname.split(".")[0] in self.roots:
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = mock.MagicMock()
            m.__path__ = []
            m.__spec__ = spec
            m.__name__ = spec.name
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, StubFinder())
    np.lib.pad = np.pad          # easy_assembler.py:329,371,424
    np.float = float             # chiron_eval.py:434, chiron_input.py:536
    from chiron import chiron_input, chiron_eval
    from chiron.utils import easy_assembler
    return chiron_input, chiron_eval, easy_assembler


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
            continue
        if r < rate:
            out.append(ch)
            out.append(rng.choice("ACGT"))
            continue
        out.append(ch)
    return "".join(out)


def overlapping_chunks(rng, n_chunks, chunk, step, rate):
    genome = rand_seq(rng, step * n_chunks + chunk)
    return [mutate(rng, genome[i * step:i * step + chunk], rate) for i in range(n_chunks)]


def main():
    ci, ce, ea = import_reference()
    rng = random.Random(20260928)
    G = {}

    # --- H2 read_signal / H3 read_data_for_eval (chiron_input.py:527-539, 253-292)
    sig_path = os.path.join(EX, "output", "raw", "read1.signal")
    sig = ci.read_signal(sig_path, normalize=None)
    G["read_signal_read1"] = {"len": len(sig), "head": sig[:16], "tail": sig[-8:],
                              "sum": float(np.sum(np.asarray(sig, dtype=np.float64)))}
    win = {}
    for (start, step, seg) in [(0, 390, 400), (0, 30, 400), (1000, 390, 400), (0, 490, 500), (61000, 400, 400)]:
        ds = ci.read_data_for_eval(sig_path, start, step, seg)
        ev, ln = ds.event, ds.event_length
        key = "%d_%d_%d" % (start, step, seg)
        win[key] = {"n": len(ev), "lengths_head": ln[:3], "lengths_tail": ln[-4:],
                    "first_head": ev[0][:6], "last": ev[-1][:8] + ev[-1][-4:],
                    "last_nonzero": int(np.count_nonzero(np.asarray(ev[-1]))),
                    "checksum": float(sum(float(np.sum(np.asarray(e, dtype=np.float64)) * (i % 7 + 1))
                                          for i, e in enumerate(ev))),
                    "reads_n": ds.reads_n}
    G["read_data_for_eval_read1"] = win

    pads = []
    for x, L in [([1.0, 2.0], 5), ([], 3), ([4.0, 5.0, 6.0], 3)]:
        y = list(x)
        ci.padding(y, L)
        pads.append({"x": x, "L": L, "out": y})
    G["padding"] = pads

    # --- P1 sparse2dense / slice_sparse_tensor / slice_ctc_decoding_result (chiron_eval.py:36-98)
    ST = ce.SparseTensor
    sp_cases = []
    for seed in range(6):
        r = np.random.RandomState(seed)
        B = int(r.randint(3, 9))
        rows = [list(map(int, r.randint(0, 4, size=r.randint(0, 5)))) for _ in range(B)]
        idx = [(b, j) for b, row in enumerate(rows) for j in range(len(row))]
        val = [v for row in rows for v in row]
        mx = max([len(row) for row in rows] + [0])
        spt = ST(indices=np.asarray(idx, dtype=np.int64).reshape(-1, 2), values=np.asarray(val, dtype=np.int64),
                 dense_shape=np.asarray([B, mx], dtype=np.int64))
        logp = r.randn(B, 1).astype(np.float32)
        reads, uniq = ce.sparse2dense(([spt], logp))
        s, e = 1, B - 1
        sl, lp = ce.slice_ctc_decoding_result(([spt], logp), s, e)
        sp_cases.append({"indices": spt.indices.tolist(), "values": spt.values.tolist(),
                         "dense_shape": spt.dense_shape.tolist(),
                         "reads": [list(map(int, x)) for x in reads[0]], "uniq": list(map(int, uniq[0])),
                         "slice": [s, e], "slice_indices": sl[0].indices.tolist(),
                         "slice_values": sl[0].values.tolist(), "slice_shape": list(map(int, sl[0].dense_shape)),
                         "logp": logp.ravel().tolist(), "slice_logp": lp.ravel().tolist()})
    G["sparse"] = sp_cases

    G["index2base"] = [{"in": x, "out": ce.index2base(x)} for x in ([0, 1, 2, 3, 3], [], [3, 3, 0])]
    G["assembler_kernal"] = [{"jump": j, "seg": l, "out": ce.get_assembler_kernal(j, l)}
                             for (j, l) in [(390, 400), (30, 400), (400, 400), (360, 400), (361, 400),
                                            (490, 500), (1900, 2000), (450, 500), (451, 500), (500, 400)]]
    G["mapping"] = [{"in": p, "out": list(map(int, ea.mapping(p)))}
                    for p in ([1, 0, 4, 3, 2, 2, 4, 2, 3], [4, 4, 4], [0, 0, 0, 1, 4, 1], [],
                              [int(v) for v in np.random.RandomState(3).randint(0, 5, 40)])]

    # --- Q1 qs (chiron_eval.py:152-174)
    qs_cases = []
    for seed in range(5):
        r = np.random.RandomState(100 + seed)
        n = int(r.randint(3, 30))
        cons = np.zeros((4, n))
        cqs = np.zeros((4, n))
        for col in range(n):
            # guarantee a strictly positive top count (n1=0 divides by zero in the reference)
            counts = r.randint(0, 6, size=4)
            counts[r.randint(0, 4)] += 1
            cons[:, col] = counts
            cqs[:, col] = counts * r.uniform(0.5, 12.0, size=4)
        qs_cases.append({"consensus": cons.tolist(), "consensus_qs": cqs.tolist(),
                         "phred": ce.qs(cons, cqs), "number": ce.qs(cons, cqs, "number").tolist()})
    hand = np.asarray([[3, 0, 1], [0, 2, 1], [0, 0, 1], [1, 0, 0]], dtype=np.float64)
    handq = np.asarray([[9, 0, 2], [0, 5, 2], [0, 0, 2], [1, 0, 0]], dtype=np.float64)
    qs_cases.append({"consensus": hand.tolist(), "consensus_qs": handq.tolist(),
                     "phred": ce.qs(hand, handq), "number": ce.qs(hand, handq, "number").tolist()})
    G["qs"] = qs_cases

    # --- A2 kernels (easy_assembler.py:212-300)
    pairs = []
    for i in range(60):
        n = rng.choice([3, 8, 12, 20, 35, 45, 60])
        prev = rand_seq(rng, n)
        ov = rng.randint(0, max(0, min(n // 2, 12)))
        cur = mutate(rng, prev[n - ov:], 0.1 if i % 3 else 0.0) + rand_seq(rng, rng.choice([0, 5, 20, 40]))
        if i == 0:
            cur, prev = "", "ACGTACGTACGT"
        if i == 1:
            cur, prev = "ACGT", ""
        if i == 2:
            cur = prev
        rec = {"cur": cur, "prev": prev, "glue": int(ea.glue_kernal(cur, prev)),
               "stick": int(ea.stick_kernal(cur, prev))}
        if len(cur) and len(prev):
            d, lp = ea.simple_assembly_kernal(cur, prev, 0.2, 0.075)
            rec["simple"] = [int(d), float(lp)]
            d2, lp2 = ea.simple_assembly_kernal(cur, prev, 0.2, 0.975)
            rec["simple_975"] = [int(d2), float(lp2)]
        pairs.append(rec)
    d, lp = ea.simple_assembly_kernal("GACCATTGACGTAC", "ACGTACGTTTGACCA", 0.2, 0.975)
    pairs.append({"cur": "GACCATTGACGTAC", "prev": "ACGTACGTTTGACCA", "simple_975": [int(d), float(lp)],
                  "glue": int(ea.glue_kernal("GACCATTGACGTAC", "ACGTACGTTTGACCA")), "stick": 15})
    G["kernels"] = pairs

    # --- A3 simple_assembly / simple_assembly_qs (easy_assembler.py:302-335,393-442)
    asm = []
    for case, (kern, n_chunks, chunk, step, rate, jr) in enumerate([
            ("glue", 12, 40, 38, 0.05, 0.975), ("glue", 60, 45, 43, 0.08, 0.975),
            ("simple", 15, 40, 6, 0.1, 0.075), ("simple", 40, 30, 4, 0.15, 0.075),
            ("stick", 8, 25, 25, 0.0, 1.0), ("glue", 3, 4, 3, 0.0, 0.975)]):
        chunks = [c for c in overlapping_chunks(rng, n_chunks, chunk, step, rate) if len(c) > 0]
        qsl = np.random.RandomState(case).uniform(0.1, 9.0, size=(len(chunks), 1))
        cons = ea.simple_assembly(chunks, jr, kernal=kern)
        cons_q, cons_qs = ea.simple_assembly_qs(chunks, qsl, jr, kernal=kern)
        assert np.array_equal(cons, cons_q)
        asm.append({"kernal": kern, "jump_ratio": jr, "chunks": chunks, "qs_list": qsl.ravel().tolist(),
                    "consensus": cons.astype(int).tolist(), "consensus_qs": cons_qs.tolist(),
                    "argmax": ce.index2base(np.argmax(cons, axis=0)), "qs_string": ce.qs(cons_q, cons_qs)})
    G["assembly"] = asm

    # --- reference example: segments -> consensus must equal result (glue, 390/400)
    ex = {}
    for i in range(1, 6):
        seg_file = os.path.join(EX, "output", "segments", "read%d.fastq" % i)
        res_file = os.path.join(EX, "output", "result", "read%d.fastq" % i)
        lines = open(seg_file).read().split("\n")
        segs = [lines[j + 1] for j in range(0, len(lines) - 1, 2) if lines[j].startswith(">")]
        cons = ea.simple_assembly(segs, 390 / 400, kernal=ce.get_assembler_kernal(390, 400))
        seq = ce.index2base(np.argmax(cons, axis=0))
        want = open(res_file).read().split("\n")[1]
        ex["read%d" % i] = {"n_segments": len(segs), "len": len(seq), "reference_matches_result": seq == want}
        assert seq == want, "reference glue assembly no longer reproduces its own example output"
    G["example_consensus"] = ex

    with open(os.path.join(HERE, "host_golden.json"), "w") as f:
        json.dump(G, f, indent=0, sort_keys=True)

    # data fixtures: the reference's own example inputs/outputs
    dst = os.path.join(HERE, "example_dna")
    for sub in ("segments", "result", "raw"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
    for i in range(1, 6):
        for sub in ("segments", "result"):
            shutil.copyfile(os.path.join(EX, "output", sub, "read%d.fastq" % i),
                            os.path.join(dst, sub, "read%d.fastq" % i))
    shutil.copyfile(sig_path, os.path.join(dst, "raw", "read1.signal"))
    shutil.copyfile(os.path.join(EX, "read1.fast5"), os.path.join(dst, "read1.fast5"))
    # checkpoint metadata of the shipped models (index = variable names/shapes/offsets; the *.data
    # blobs are stripped from the reference tree): model folders usable with --synthetic-weights
    pkg_models = os.path.join(os.path.dirname(os.path.dirname(HERE)), "chiron_amd", "model")
    for m in ("DNA_default", "RNA_default"):
        src = os.path.join(REF, "chiron", "model", m)
        os.makedirs(os.path.join(pkg_models, m), exist_ok=True)
        for fn in os.listdir(src):
            if fn.endswith(".meta"):
                continue
            shutil.copyfile(os.path.join(src, fn), os.path.join(pkg_models, m, fn))
            os.chmod(os.path.join(pkg_models, m, fn), 0o644)
    for p, _, fs in os.walk(dst):
        for fn in fs:
            os.chmod(os.path.join(p, fn), 0o644)
    print("wrote", os.path.join(HERE, "host_golden.json"))


if __name__ == "__main__":
    main()
