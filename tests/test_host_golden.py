"""CPU: the host-side pipeline functions against golden vectors captured from the reference's own
Python functions (tests/golden/make_golden.py) and against the reference's checked-in example
outputs.  Rows refer to SURVEY.md section 8(a)."""
import os

import numpy as np
import pytest

from chiron_amd import assembly, entry, signal_io
from chiron_amd import eval as ce
from conftest import GOLDEN

EX = os.path.join(GOLDEN, "example_dna")
SIG1 = os.path.join(EX, "raw", "read1.signal")


def test_read_signal_H2(golden):
    g = golden["read_signal_read1"]
    sig = signal_io.read_signal(SIG1)
    assert sig.dtype == np.float32 and len(sig) == g["len"] == 62461
    assert sig[:16].tolist() == g["head"] and sig[-8:].tolist() == g["tail"]
    assert float(np.sum(sig.astype(np.float64))) == g["sum"]


@pytest.mark.parametrize("key", ["0_390_400", "0_30_400", "1000_390_400", "0_490_500", "61000_400_400"])
def test_read_data_for_eval_H3(golden, key):
    g = golden["read_data_for_eval_read1"][key]
    start, step, seg = map(int, key.split("_"))
    ds = signal_io.read_data_for_eval(SIG1, start, step, seg)
    ev, ln = ds.event, ds.event_length
    assert ds.reads_n == g["n"] == g["reads_n"]
    assert ln[:3].tolist() == g["lengths_head"] and ln[-4:].tolist() == g["lengths_tail"]
    assert ev[0][:6].tolist() == g["first_head"]
    assert ev[-1][:8].tolist() + ev[-1][-4:].tolist() == g["last"]
    assert int(np.count_nonzero(ev[-1])) == g["last_nonzero"]
    chk = float(sum(float(np.sum(e.astype(np.float64)) * (i % 7 + 1)) for i, e in enumerate(ev)))
    assert chk == g["checksum"]


def test_read_data_for_eval_rejects_other_suffix():
    with pytest.raises(TypeError):
        signal_io.read_data_for_eval("foo.txt")


def test_empty_and_short_signals(tmp_path):
    p = tmp_path / "e.signal"
    p.write_text("")
    ds = signal_io.read_data_for_eval(str(p), 0, 390, 400)
    assert ds.reads_n == 0
    p.write_text("5 6\n7")
    ds = signal_io.read_data_for_eval(str(p), 0, 390, 400)
    assert ds.reads_n == 1 and ds.event_length.tolist() == [3] and ds.event[0, :4].tolist() == [5, 6, 7, 0]


def test_next_batch_H4():
    ev = np.arange(10 * 4, dtype=np.float32).reshape(10, 4)
    ds = signal_io.DataSet(ev, np.full(10, 4))
    x, l, _ = ds.next_batch(4)
    assert x.shape == (4, 4) and ds.epochs_completed == 0
    x, l, _ = ds.next_batch(4)
    assert ds.epochs_completed == 0
    x, l, _ = ds.next_batch(4)
    assert x.shape == (2, 4) and ds.epochs_completed == 1 and l.dtype == np.int32
    ds = signal_io.DataSet(ev, np.full(10, 4))
    x, _, _ = ds.next_batch(10)                    # exactly the remainder also completes the epoch
    assert x.shape == (10, 4) and ds.epochs_completed == 1


def test_batch_packer_H5_toy_stream():
    """Hand-derived table: reads of 3/9/2 windows, B=4, L=2, ratio 1."""
    L, B = 2, 4
    reads = [("a", 3), ("b", 9), ("c", 2)]
    packer = ce.BatchPacker(B, L, 1.0)
    batches = []
    for name, n in reads:
        ev = np.arange(n * L, dtype=np.float32).reshape(n, L) + (ord(name) - 96) * 100
        ln = np.full(n, L, dtype=np.int32)
        ln[-1] = 1
        batches += list(packer.add_read(name, ev, ln))
    last = packer.flush()
    assert last is not None
    batches.append(last)
    assert [b.n_valid for b in batches] == [4, 4, 4, 2]
    assert batches[0].fname.tolist() == ["a", "a", "a", "b"] and batches[0].index.tolist() == [0, 0, 0, 0]
    assert batches[1].fname.tolist() == ["b"] * 4 and batches[1].index.tolist() == [1, 1, 1, 1]
    assert batches[2].fname.tolist() == ["b"] * 4 and batches[2].index.tolist() == [5, 5, 5, 5]
    assert batches[3].fname.tolist() == ["c", "c", "", ""] and batches[3].index.tolist() == [0, 0, -1, -1]
    # wrap padding repeats the real rows (np.pad mode='wrap', chiron_eval.py:352-360)
    assert np.array_equal(batches[3].x[2], batches[3].x[0]) and np.array_equal(batches[3].x[3], batches[3].x[1])
    assert batches[3].seq_len.tolist() == [2, 1, 2, 1]
    assert batches[0].seq_len.tolist() == [2, 2, 1, 2] and batches[0].seq_len.dtype == np.int32
    # the pipeline walks a batch's per-read runs; the reference's per-row tags above are their expansion, and back
    assert batches[0].runs == [("a", 0, 3, 0), ("b", 3, 1, 0)] and batches[3].runs == [("c", 0, 2, 0)]
    for b in batches:
        again = ce.Batch.from_tags(b.x, b.seq_len, b.fname, b.index, b.n_valid)
        assert again.runs == b.runs and again.fname.tolist() == b.fname.tolist() and again.index.tolist() == b.index.tolist()


def test_seq_len_round_half_even():
    from chiron_amd import seq_len_for_engine
    # RNA shipped ratio 5: 2.5 -> 2, 7.5 -> 8 (np.round), chiron_eval.py:337
    assert seq_len_for_engine([12, 13, 37, 38, 500, 0], 5.0).tolist() == [2, 3, 7, 8, 100, 0]


def test_sparse_P1(golden):
    for c in golden["sparse"]:
        sp = ce.SparseTensor(np.asarray(c["indices"], dtype=np.int64).reshape(-1, 2), np.asarray(c["values"], dtype=np.int64),
                             np.asarray(c["dense_shape"], dtype=np.int64))
        logp = np.asarray(c["logp"], dtype=np.float32).reshape(-1, 1)
        reads, uniq = ce.sparse2dense(([sp], logp))
        assert [list(map(int, r)) for r in reads[0]] == c["reads"]
        assert list(map(int, uniq[0])) == c["uniq"]
        s, e = c["slice"]
        sl, lp = ce.slice_ctc_decoding_result(([sp], logp), s, e)
        assert sl[0].indices.tolist() == c["slice_indices"] and sl[0].values.tolist() == c["slice_values"]
        assert list(map(int, sl[0].dense_shape)) == c["slice_shape"]
        assert lp.ravel().tolist() == c["slice_logp"]


def test_index2base_kernal_mapping_P2_A1(golden):
    for c in golden["index2base"]:
        assert ce.index2base(c["in"]) == c["out"]
    for c in golden["assembler_kernal"]:
        assert ce.get_assembler_kernal(c["jump"], c["seg"]) == c["out"]
    for c in golden["mapping"]:
        assert list(map(int, assembly.mapping(c["in"]))) == c["out"]


def test_qs_Q1(golden):
    """Reference outputs of qs() (chiron_eval.py:152-174).  Where a column's two largest counts are EQUAL the reference
    reads the quality sum of whichever tied base np.argsort happens to place last -- an implementation detail of the
    NumPy build (insertion sort in NumPy 1.x: the higher base index; the SIMD sorting network of NumPy 2 on this
    container's CPU: not stable).  Untied columns must match the captured output exactly; a tied column must match one
    of the tied bases' values, and the product's own rule for it (highest base index, i.e. NumPy 1.x) is pinned below."""
    tied_seen = 0
    for c in golden["qs"]:
        cons, cqs = np.asarray(c["consensus"], dtype=float), np.asarray(c["consensus_qs"], dtype=float)
        tied_seen += _assert_qs_matches_reference(cons, cqs, c["phred"])
        assert ce.qs(cons, cqs, "number").tolist() == [ord(ch) - 33 for ch in ce.qs(cons, cqs)]
    assert tied_seen > 0


def _assert_qs_matches_reference(cons, cqs, ref_string):
    """-> number of tied-top columns met"""
    got = ce.qs(cons, cqs)
    assert len(got) == len(ref_string)
    srt = np.sort(cons, axis=0)
    tied = np.flatnonzero(srt[3] == srt[2])
    keep = np.ones(len(got), dtype=bool)
    keep[tied] = False
    assert "".join(np.array(list(got))[keep]) == "".join(np.array(list(ref_string))[keep])
    for col in tied:
        n1 = srt[3, col]
        options = {int(cqs[b, col] / n1 / np.log(10)) for b in range(4) if cons[b, col] == n1}
        hi = max(b for b in range(4) if cons[b, col] == n1)
        assert ord(ref_string[col]) - 33 in options
        assert ord(got[col]) - 33 == int(cqs[hi, col] / n1 / np.log(10))
    return len(tied)


def test_qs_ties_follow_the_stable_argsort_of_the_references_numpy():
    """Q16: chiron_eval.py:165-170 reads count and quality sum at position 3 of np.argsort(consensus, axis=0).  Among EQUAL top
    counts the base that lands there depends on the sort: NumPy 1.x (the reference's era) sorts four elements with an
    insertion sort, i.e. stably -- the highest base index is last; NumPy 2's SIMD sorting network on this container's CPU is
    not stable.  The product pins the stable rule (eval.qs, consensus.hip).  The formula below is the reference's, restated
    with kind='stable'; matrices are built to contain two-, three- and four-way ties at the top and at the second rank."""
    from chiron_amd import eval as ev
    rng = np.random.RandomState(11)
    for trial in range(300):
        n = rng.randint(1, 40)
        counts = rng.randint(0, 4, size=(4, n)).astype(np.float64)          # small range: ties in most columns
        counts[rng.randint(0, 4), :] += 1.0                                  # no all-zero column (Q15 is tested on its own)
        if trial % 3 == 0:
            counts[:, : n // 2] = counts[0:1, : n // 2] + 1.0                # four-way ties, never all zero
        q_sum = np.round(rng.uniform(0, 9, size=(4, n)) * counts, 3)
        sort_ind = np.argsort(counts, axis=0, kind="stable")
        cols = np.arange(n)[np.newaxis, :]
        sc, sq = counts[sort_ind, cols], q_sum[sort_ind, cols]
        want = (10 * np.log10((sc[3] + 1) / (sc[2] + 1)) + sq[3] / sc[3] / np.log(10)).astype(int)
        assert np.array_equal(ev.qs(counts, q_sum, "number"), want), trial
        assert ev.qs(counts, q_sum) == "".join(chr(v + 33) for v in want)
    # and the shared formula on a vote summary (the device vote's path) gives the same characters
    counts = np.array([[3., 3., 0.], [3., 1., 2.], [0., 3., 2.], [1., 0., 2.]])
    q_sum = np.array([[9., 6., 0.], [7.5, 1., 4.], [0., 5.25, 4.5], [1., 0., 5.]])
    assert ev.qs(counts, q_sum) == ev.qs_from_votes(np.array([3., 3., 2.]), np.array([3., 3., 2.]), np.array([7.5, 5.25, 5.]))


def test_qs_zero_vote_column_and_shapes():
    """a column with no votes (n1 = 0) makes the reference divide 0 by 0 and chr() a garbage int
    (chiron_eval.py:168-169); decision: such a column scores 0 ('!') and the rest of the read is unaffected."""
    cons = np.array([[3., 0., 0.], [0., 0., 2.], [1., 0., 2.], [0., 0., 0.]])
    cqs = np.array([[9., 0., 0.], [0., 0., 5.], [2., 0., 7.], [0., 0., 0.]])
    assert ce.qs(cons, cqs, "number").tolist() == [int(10 * np.log10(4 / 2) + 9 / 3 / np.log(10)), 0, int(7 / 2 / np.log(10))]
    assert ce.qs(cons, cqs)[1] == "!"
    assert ce.qs(np.zeros((4, 0)), np.zeros((4, 0))) == "" and ce.qs(np.zeros((4, 0)), np.zeros((4, 0)), "number").shape == (0,)
    with pytest.raises(ValueError):
        ce.qs(cons, cqs, "phred+64")


def test_kernels_A2(golden, built):
    for p in golden["kernels"]:
        assert assembly.glue_kernal(p["cur"], p["prev"]) == p["glue"]
        assert assembly.stick_kernal(p["cur"], p["prev"]) == p["stick"]
        for key, jr in (("simple", 0.075), ("simple_975", 0.975)):
            if key in p:
                d, lp = assembly.simple_assembly_kernal(p["cur"], p["prev"], 0.2, jr)
                assert [d, lp] == p[key]
        # the native (C++) glue agrees with the Python form on every pair
        if p["cur"] and p["prev"]:
            cons = assembly.simple_assembly([p["prev"], p["cur"]], 0.975, kernal="glue")
            assert cons.shape[1] == p["glue"] + len(p["cur"])


def test_assembly_A3(golden, built):
    for c in golden["assembly"]:
        qsl = np.asarray(c["qs_list"]).reshape(-1, 1)
        cons = assembly.simple_assembly(c["chunks"], c["jump_ratio"], kernal=c["kernal"])
        cq, cqs = assembly.simple_assembly_qs(c["chunks"], qsl, c["jump_ratio"], kernal=c["kernal"])
        assert np.array_equal(cons, np.asarray(c["consensus"], dtype=np.float64))
        assert np.array_equal(cq, cons)
        np.testing.assert_allclose(cqs, np.asarray(c["consensus_qs"]), rtol=1e-12, atol=0)
        assert ce.index2base(np.argmax(cons, axis=0)) == c["argmax"]
        _assert_qs_matches_reference(cq, cqs, c["qs_string"])


def test_assembly_quirks(built):
    """single segment -> empty consensus (the reference's `continue` skips the length update);
    no segments -> empty."""
    assert assembly.simple_assembly(["ACGT"], 0.975, kernal="glue").shape == (4, 0)
    assert assembly.simple_assembly([], 0.975, kernal="glue").shape == (4, 0)
    c = assembly.simple_assembly(["ACGTACGTACGTACGTACGTACGT", "A"], 0.975, kernal="stick")
    assert c.shape == (4, 25)


@pytest.mark.parametrize("i", [1, 2, 3, 4, 5])
def test_example_reads_segments_to_result(golden, built, i):
    """The reference's checked-in example: segments/readN.fastq -> result/readN.fastq sequence line, exact
    (glue, jump 390 / segment 400).  Pins rows A1-A3 end to end."""
    lines = open(os.path.join(EX, "segments", "read%d.fastq" % i)).read().split("\n")
    segs = [lines[j + 1] for j in range(0, len(lines) - 1, 2) if lines[j].startswith(">")]
    assert len(segs) == golden["example_consensus"]["read%d" % i]["n_segments"]
    cons = assembly.simple_assembly(segs, 390 / 400, kernal=ce.get_assembler_kernal(390, 400))
    seq = ce.index2base(np.argmax(cons, axis=0))
    want = open(os.path.join(EX, "result", "read%d.fastq" % i)).read().split("\n")[1]
    assert seq == want and len(seq) == golden["example_consensus"]["read%d" % i]["len"]


def test_write_output_O1(tmp_path):
    class F(object):
        output = str(tmp_path)
        mode = "dna"
        batch_size, segment_len, jump, start = 400, 400, 390, 0
        input, model = "in/", "model/"
    import time
    t0 = time.time()
    ce.write_output(["ACG", "TT"], "ACGTT", [t0, 0.1, 0.2, 0.3], "read1", F, suffix="fastq", q_score="!!!!!")
    assert open(os.path.join(str(tmp_path), "result", "read1.fastq")).read() == "@read1\nACGTT\n+\n!!!!!\n"
    assert open(os.path.join(str(tmp_path), "segments", "read1.fastq")).read() == ">read10\nACG\n>read11\nTT\n"
    meta = open(os.path.join(str(tmp_path), "meta", "read1.meta")).read().split("\n")
    assert meta[0] == "# Reading Basecalling assembly output total rate(bp/s)"
    assert meta[2] == "# read_len batch_size segment_len jump start_pos" and meta[3] == "5 400 400 390 0"
    assert meta[5] == "in/ model/"
    F.mode = "rna"
    ce.write_output(["ACG"], "ACGTT", [t0, 0.1, 0.2, 0.3], "r2", F, suffix="fasta", concise=True)
    assert open(os.path.join(str(tmp_path), "result", "r2.fasta")).read() == ">r2\nACGUU"


def test_presets_E0():
    p = entry.build_parser()
    a = entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y"]))
    assert (a.batch_size, a.segment_len, a.jump, a.beam, a.start) == (400, 500, 490, 30, 0)
    a = entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y", "-p", "dna-pre", "-b", "1100", "--beam", "0"]))
    assert (a.batch_size, a.segment_len, a.jump, a.beam) == (1100, 400, 390, 0)
    a = entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y", "-p", "rna-pre", "--mode", "rna"]))
    assert (a.batch_size, a.segment_len, a.jump) == (300, 2000, 1900)
    with pytest.raises(ValueError):
        entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y", "-p", "dna-pre", "--mode", "rna"]))
    with pytest.raises(ValueError):
        entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y", "-p", "rna-pre"]))
    with pytest.raises(ValueError):
        entry.resolve_preset(p.parse_args(["call", "-i", "x", "-o", "y", "-p", "bogus"]))


def test_read_collector_orders_pieces_by_window_index():
    """A read spanning 3 batches is re-assembled in within-file order even when batches are drained
    out of order (intended semantic)."""
    from chiron_amd.engine import DecodeResult, SparseTensor
    col = ce.ReadCollector()
    col.expect("b", 9, (0.0, 0.0))

    def mk(batch_rows, first_idx, base):
        fn = np.asarray(["b"] * batch_rows, dtype=object)
        idx = np.full(batch_rows, first_idx)
        ind = np.asarray([[r, 0] for r in range(batch_rows)], dtype=np.int64)
        val = np.asarray([(base + r) % 4 for r in range(batch_rows)], dtype=np.int64)
        res = DecodeResult(SparseTensor(ind, val, np.asarray([batch_rows, 1])), np.zeros((batch_rows, 1), np.float32),
                           np.arange(batch_rows, dtype=np.float32).reshape(-1, 1) + base, None)
        return ce.Batch.from_tags(None, None, fn, idx, batch_rows), res
    b2, r2 = mk(4, 5, 5)
    b0, r0 = mk(1, 0, 0)
    b1, r1 = mk(4, 1, 1)
    assert col.add_batch(b2, r2, True) == []
    assert col.add_batch(b0, r0, True) == []
    done = col.add_batch(b1, r1, True)
    assert len(done) == 1
    name, flat, seg_len, qs_list, meta = done[0]
    reads = ce.split_flat(flat, seg_len)
    assert flat.dtype == np.uint8 and seg_len.tolist() == [1] * 9
    assert [int(r[0]) for r in reads] == [i % 4 for i in range(9)]
    assert qs_list.ravel().tolist() == list(map(float, range(9)))


def test_native_signal_text_parser_equals_numpy(built, tmp_path):
    """chiron_parse_signal_text (C ABI) vs the numpy conversion of chiron_input.py:527-539 on integers, floats,
    exponents, mixed whitespace, empty input; a bad token raises like the reference's float()."""
    import ctypes as C
    from chiron_amd import _lib, signal_io
    rng = np.random.RandomState(5)
    ints = " ".join(str(int(v)) for v in rng.randint(-2000, 2000, size=5000))
    floats = "\n".join(repr(float(v)) for v in rng.randn(3000) * 1e3) + "\t1e-3  -2.5E+4 +7 .5 5. nan inf\r\n"
    for text in (ints, floats, "", "   \n", "42", ints + "\n" + floats):
        p = tmp_path / "t.signal"
        p.write_text(text)
        got = signal_io.read_signal(str(p))
        want = np.asarray(text.split(), dtype=np.float32)
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True)
    for bad in ("1 2 x3 4", "1 0x1p3 2", "0X10", "1 2e 3", "--4"):      # incl. C99 hex floats, which strtod would accept
        p.write_text(bad)
        with pytest.raises(ValueError, match="could not convert"):
            signal_io.read_signal(str(p))
        with pytest.raises(ValueError):
            np.asarray(bad.split(), dtype=np.float32)                  # the reference's conversion refuses them too
    lib = _lib.load()
    out = np.empty(2, np.float32)
    n = C.c_size_t()
    assert lib.chiron_parse_signal_text(b"1 2 3", 5, out.ctypes.data_as(C.c_void_p), 2, C.byref(n)) == _lib.ERR_OVERFLOW


def test_packer_collector_round_trip_random_streams():
    """Property (rows H5 + P1, chiron_eval.py:321-360 / :403-446): for random reads, batch sizes and drain orders every
    window comes back exactly once, in within-file order, under its own read; rows whose decode is empty vanish;
    wrap-padding rows of the last batch are never attributed to a read."""
    from chiron_amd.engine import DecodeResult, SparseTensor, CompactDecode
    rng = np.random.RandomState(77)
    for trial in range(60):
        B = int(rng.choice([1, 2, 3, 4, 7, 16]))
        L = 3
        reads = [("r%02d" % i, int(rng.randint(1, 20))) for i in range(int(rng.randint(1, 9)))]
        packer, col = ce.BatchPacker(B, L, 1.0), ce.ReadCollector()
        batches, token = [], {}
        uid = 0
        for name, n in reads:
            ev = np.zeros((n, L), dtype=np.float32)
            for k in range(n):
                ev[k, 0] = uid      # a unique id per window travels in the signal
                token[uid] = (name, k)
                uid += 1
            col.expect(name, n, (0.0, 0.0))
            batches += list(packer.add_read(name, ev, np.full(n, L, dtype=np.int32)))
        last = packer.flush()
        if last is not None:
            batches.append(last)
        assert sum(b.n_valid for b in batches) == uid and all(b.x.shape == (B, L) for b in batches)
        # "decode": window uid -> (uid % 3) labels (so every third window decodes to nothing), qs = uid
        order = rng.permutation(len(batches))
        done, done_c = [], []
        col_c = ce.ReadCollector()
        for name, n in reads:
            col_c.expect(name, n, (0.0, 0.0))
        # full batches travel as the per-run pieces (Engine.submit_pieces); x is their concatenation, built on demand
        for b in batches:
            if b.pieces is not None:
                assert b.n_valid == B and sum(len(p_) for p_ in b.pieces) == B and np.array_equal(np.concatenate(b.pieces), b.x)
        for bi in order:
            b = batches[bi]
            ind, val = [], []
            for r in range(B):
                u = int(b.x[r, 0])
                for p in range(u % 3):
                    ind.append([r, p])
                    val.append((u + p) % 4)
            st = SparseTensor(np.asarray(ind, dtype=np.int64).reshape(-1, 2), np.asarray(val, dtype=np.int64), np.asarray([B, 2]))
            res = DecodeResult(st, np.zeros((B, 1), np.float32), b.x[:, :1].copy(), None)
            done += col.add_batch(b, res, True)
            # the engine's compact form of the same decode (CHIRON_COMPACT_DECODE: rows' labels back to back + labels per row) through
            # the collector's fast path: the same reads, bit for bit
            counts = np.bincount(st.indices[:, 0], minlength=B).astype(np.int32) if len(val) else np.zeros(B, np.int32)
            resc = DecodeResult(None, res.log_prob, res.prob_logits, None, CompactDecode(np.asarray(val, dtype=np.uint8), counts, np.asarray([B, 2])))
            done_c += col_c.add_batch(b, resc, True)
        assert [d[0] for d in done_c] == [d[0] for d in done]
        for a, c in zip(done, done_c):
            assert np.array_equal(a[1], c[1]) and a[1].dtype == c[1].dtype and np.array_equal(a[2], c[2]) and a[2].dtype == c[2].dtype
            assert np.array_equal(a[3], c[3]) and a[3].shape == c[3].shape
        assert sorted(d[0] for d in done) == sorted(n for n, _ in reads)
        for name, flat, seg_len, qs_list, meta in done:
            out_reads = ce.split_flat(flat, seg_len)
            n = dict(reads)[name]
            want = [u for u, (nm, k) in sorted(token.items(), key=lambda kv: kv[1][1]) if nm == name and u % 3 > 0]
            assert len(out_reads) == len(want), (trial, name)
            assert qs_list.ravel().tolist() == [float(u) for u in want]
            for rd, u in zip(out_reads, want):
                assert [int(v) for v in rd] == [(u + p) % 4 for p in range(u % 3)]


def test_native_finisher_writes_the_same_files_as_the_python_writers(built, tmp_path):
    """chiron_finish_read (one GIL-free call: index2base, vote, argmax, qs, result/ + segments/ files) against finish_read
    (the Python writers, chiron_eval.py:446-462 / :176-242) on random reads: FASTQ and FASTA, --concise, RNA (T -> U in the
    consensus only), the glue / stick / simple kernels (chosen by jump : segment_len as get_assembler_kernal does), reads
    of 0 / 1 / 2 windows, empty decodes inside a read, tied votes and quality sums that land on integer boundaries.  Every
    result / segments file must be identical byte for byte, and the returned consensus too."""
    rng = np.random.RandomState(123)

    def flags(out, ext, concise, mode, jump, python_finish):
        class F(object):
            pass
        F.output, F.extension, F.concise, F.mode, F.jump, F.segment_len = out, ext, concise, mode, jump, 400
        F.batch_size, F.start, F.input, F.model, F.python_finish = 100, 0, "in/", "m", python_finish
        for sub in ("result", "segments", "meta"):
            os.makedirs(os.path.join(out, sub), exist_ok=True)
        return F
    case = 0
    for ext in ("fastq", "fasta"):
        for concise in (False, True):
            for mode in ("dna", "rna"):
                for jump in (390, 400, 30):                  # glue, stick, simple
                    for n_seg in (0, 1, 2, 3, 40):
                        case += 1
                        base_len = 8 if jump == 30 else 30
                        truth = rng.randint(0, 4, size=n_seg * 6 + base_len + 8)
                        segs = []
                        for k in range(n_seg):               # overlapping windows of one underlying sequence, with errors
                            a = 6 * k if jump != 400 else base_len * k % max(1, truth.size - base_len)
                            sg = truth[a:a + base_len].copy()
                            flip = rng.rand(sg.size) < 0.1
                            sg[flip] = rng.randint(0, 4, size=int(flip.sum()))
                            segs.append(sg[: rng.randint(1, sg.size + 1)] if rng.rand() < 0.2 else sg)
                        seg_len = np.asarray([len(s_) for s_ in segs], dtype=np.int64)
                        flat = np.concatenate(segs).astype(np.uint8) if segs else np.zeros(0, dtype=np.uint8)
                        qs_list = np.round(rng.uniform(0, 6, size=(n_seg, 1)), rng.randint(0, 3))     # round numbers: integer boundaries
                        outs = []
                        for python_finish in (True, False):
                            F = flags(str(tmp_path / ("c%d_%d" % (case, python_finish))), ext, concise, mode, jump, python_finish)
                            cons = ce.finish_read_flat("read_x.signal", flat, seg_len, qs_list, F, 0.0, 0.0)
                            outs.append((F.output, cons))
                        assert outs[0][1] == outs[1][1], case
                        for sub in ("result",) + (() if concise else ("segments",)):
                            a = open(os.path.join(outs[0][0], sub, "read_x." + ext), "rb").read()
                            b = open(os.path.join(outs[1][0], sub, "read_x." + ext), "rb").read()
                            assert a == b, (case, sub, ext, concise, mode, jump, n_seg)
                        assert os.path.exists(os.path.join(outs[1][0], "meta", "read_x.meta")) != concise
                        if concise:
                            assert os.listdir(os.path.join(outs[1][0], "segments")) == []
