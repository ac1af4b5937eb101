"""Basecalling pipeline (counterpart of chiron/chiron_eval.py).

TensorFlow's session / FIFOQueues / QueueRunner threads are replaced by one Engine per GPU with
several batches in flight on HIP streams; the host keeps the reference's structure: cross-read
batch packing (_worker_fn, chiron_eval.py:304-368), per-file regroup (evaluation, :383-446),
consensus (:447-457) and the writers (:176-242).

Deliberate divergences from HEAD, each listed in SURVEY.md appendix D:
  Q3 regroup orders a read's pieces by within-file window index (HEAD's key shadowing scrambles
     multi-batch reads); Q4 no debug prints; Q5 each file is parsed once; Q6 sorted file order.
"""
import collections
import os
from concurrent.futures import ThreadPoolExecutor
import sys
import time

import numpy as np

from . import assembly
from . import model as model_mod
from . import signal_io
from .engine import Engine, SparseTensor, seq_len_for_engine, piece_ok
from .unix_time import unix_time

BASES = "ACGT"
DEVICE_VOTE_MIN_SEGMENTS = 8192      # reads with at least this many decoded windows take the device-side consensus vote


def _rows_of(sparse):
    """Row boundaries of a row-major sorted SparseTensor -> (row ids that occur, start offset of each, end offsets)."""
    rows = sparse.indices[:, 0]
    if rows.shape[0] == 0:
        empty = np.zeros(0, dtype=np.int64)
        return empty, empty, empty
    starts = np.concatenate(([0], np.flatnonzero(rows[1:] != rows[:-1]) + 1))
    return rows[starts], starts, np.concatenate((starts[1:], [rows.shape[0]]))


def sparse2dense(predict_val):
    """chiron_eval.py:36-66: (decoded SparseTensors, log_prob) -> per decoder output the ragged list of reads (one
    array of base indices per row that decoded to something) and the ids of those rows.  Rows with an empty decode do
    not appear in a SparseTensor, hence not here either."""
    predict_read, uniq_list = [], []
    for decode in predict_val[0]:
        ids, starts, ends = _rows_of(decode)
        predict_read.append([decode.values[a:b] for a, b in zip(starts, ends)])
        uniq_list.append(ids)
    return predict_read, uniq_list


def slice_sparse_tensor(input_sp, start, end):
    """chiron_eval.py:68-83: the rows [start, end) of a SparseTensor, renumbered from 0 (same dense width).  The
    decoders emit indices sorted by row, so the slice is one contiguous range found by bisection."""
    rows = input_sp.indices[:, 0]
    lo, hi = np.searchsorted(rows, [start, end], side="left")
    indices = input_sp.indices[lo:hi].copy()
    indices[:, 0] -= start
    return SparseTensor(indices=indices, values=input_sp.values[lo:hi],
                        dense_shape=np.asarray([end - start, input_sp.dense_shape[1]]))


def slice_ctc_decoding_result(input_decode, start, end):
    """chiron_eval.py:85-98: slice every decoder output and the [batch, 1] log-probabilities alike."""
    decoded, log_prob = input_decode
    return [slice_sparse_tensor(d, start, end) for d in decoded], log_prob[start:end, :]


_BASE_CODES = np.frombuffer(BASES.encode("ascii"), dtype=np.uint8)


def index2base(read):
    """chiron_eval.py:100-113: base indices -> string, as one table lookup (an index outside 0..3 raises IndexError as
    the reference's per-element lookup does; negative indices count from the end, likewise)."""
    idx = np.asarray(read)
    if idx.size == 0:
        return ""
    return _BASE_CODES[idx.astype(np.intp, copy=False)].tobytes().decode("ascii")


def bases_of_reads(reads):
    """index2base over a ragged list of reads with one lookup for all of them (a long read has hundreds of windows)."""
    if len(reads) == 0:
        return []
    lengths = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    if lengths.sum() == 0:
        return [""] * len(reads)
    text = index2base(np.concatenate([np.asarray(r).ravel() for r in reads]))
    ends = np.cumsum(lengths)
    return [text[e - n:e] for e, n in zip(ends.tolist(), lengths.tolist())]


def get_assembler_kernal(jump, segment_len):
    """chiron_eval.py:138-150."""
    assembler = "simple"
    if jump > 0.9 * segment_len:
        assembler = "glue"
    if jump >= segment_len:
        assembler = "stick"
    return assembler


def qs(consensus, consensus_qs, output_standard="phred+33"):
    """chiron_eval.py:152-174: per consensus column, with n1 >= n2 the two largest vote counts and Q the summed segment
    quality behind the winning base:   q = 10 log10((n1 + 1) / (n2 + 1)) + (Q / n1) / ln 10,   truncated to int.
    Among equal top counts the reference's ascending argsort leaves the highest base index last, i.e. that base's Q is
    used; kept.  A column nobody voted for (n1 = 0; the reference would divide 0 by 0 and fail in chr()) gets q = 0."""
    counts = np.asarray(consensus, dtype=np.float64)
    n_col = counts.shape[1]
    if n_col == 0:
        return np.zeros(0, dtype=int) if output_standard == "number" else ""
    top = counts.shape[0] - 1 - np.argmax(counts[::-1], axis=0)      # last index among the maxima
    n1 = np.take_along_axis(counts, top[None, :], axis=0)[0]
    rest = counts.copy()                                             # second largest count: the maximum without the winner
    np.put_along_axis(rest, top[None, :], -1.0, axis=0)
    n2 = rest.max(axis=0)
    q_top = np.take_along_axis(np.asarray(consensus_qs, dtype=np.float64), top[None, :], axis=0)[0]
    return qs_from_votes(n1, n2, q_top, output_standard)


def qs_from_votes(n1, n2, q_top, output_standard="phred+33"):
    """The formula of qs() on a column's vote summary (float64 arrays): n1 >= n2 the two largest counts, q_top the quality
    sum behind the winning base.  Shared by the host vote (qs) and the device vote (assembly.consensus_device), so both
    produce the same characters whatever log10 the device has."""
    n_col = n1.shape[0]
    if n_col == 0:
        return np.zeros(0, dtype=int) if output_standard == "number" else ""
    voted = n1 > 0
    if voted.all():
        score = 10 * np.log10((n1 + 1) / (n2 + 1)) + q_top / n1 / np.log(10)
    else:
        score = np.zeros(n_col)
        score[voted] = 10 * np.log10((n1[voted] + 1) / (n2[voted] + 1)) + q_top[voted] / n1[voted] / np.log(10)
    score = score.astype(int)
    if output_standard == "number":
        return score
    if output_standard == "phred+33":
        codes = score + 33
        if codes.min() >= 0 and codes.max() < 128:      # every realistic score: one ASCII byte per column
            return codes.astype(np.uint8).tobytes().decode("ascii")
        return "".join(map(chr, codes))                 # as the reference: chr() of a large code, ValueError for a negative one
    raise ValueError("output_standard must be 'number' or 'phred+33'")


class OutputTree(object):
    """The folders `chiron call` fills (README.md:154-160): result/ (one consensus per read), segments/ (one record per
    window), meta/ (timings).  One writer per file type; chiron_eval.py:176-242 holds the formats."""

    def __init__(self, root, suffix="fasta", concise=False, rna=False):
        self.root, self.suffix, self.concise, self.rna = root, suffix, concise, rna

    def _path(self, folder, file_pre, ext):
        return os.path.join(self.root, folder, file_pre + "." + ext)

    @staticmethod
    def _write(path, text):
        """One write per file; the folder is created on demand (a recursive input puts sub-folders into file_pre)."""
        try:
            f = open(path, "w")
        except FileNotFoundError:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            f = open(path, "w")
        with f:
            f.write(text)

    def consensus(self, file_pre, sequence, quality=None):
        """result/<pre>.fastq = @name / sequence / + / quality (each line terminated); result/<pre>.fasta = >name /
        sequence WITHOUT a final newline (chiron_eval.py:216-221)."""
        if self.rna:                                          # chiron_eval.py:204-205
            sequence = sequence.replace("T", "U").replace("t", "u")
        if self.suffix == "fastq" and quality is not None:
            text = "@%s\n%s\n+\n%s\n" % (file_pre, sequence, quality)
        else:
            text = ">%s\n%s" % (file_pre, sequence)
        self._write(self._path("result", file_pre, self.suffix), text)
        return sequence

    def segments(self, file_pre, reads, qualities=None):
        """segments/<pre>.<suffix>: a FASTA-style record per window, named <pre><window index>; with per-window quality
        strings (never passed by `chiron call`, chiron_eval.py:460-462) a FASTQ record follows each."""
        records = []
        for k, read in enumerate(reads):
            records.append(">%s%d\n%s\n" % (file_pre, k, read))
            if self.suffix == "fastq" and qualities is not None:
                records.append("@%s%d\n%s\n+\n%s\n" % (file_pre, k, read, qualities[k]))
        self._write(self._path("segments", file_pre, self.suffix), "".join(records))

    def meta(self, file_pre, n_bases, stamps, settings):
        """meta/<pre>.meta: stage durations derived from the cumulative stamps (start, reading, basecall, assembly) and
        the run's settings, three header/value line pairs (chiron_eval.py:229-242)."""
        start, reading, basecall_end, assembly_end = stamps
        total = time.time() - start
        spans = (reading, basecall_end - reading, assembly_end - basecall_end, total - assembly_end, total)
        self._write(self._path("meta", file_pre, "meta"),
                    "# Reading Basecalling assembly output total rate(bp/s)\n"
                    + " ".join("%5.3f" % v for v in spans + (n_bases / total,)) + "\n"
                    + "# read_len batch_size segment_len jump start_pos\n"
                    + "%d %d %d %d %d\n" % (n_bases, settings.batch_size, settings.segment_len, settings.jump, settings.start)
                    + "# input_name model_name\n"
                    + "%s %s\n" % (settings.input, settings.model))


def write_output(segments, consensus, time_list, file_pre, global_setting, concise=False, suffix="fasta",
                 seg_q_score=None, q_score=None):
    """chiron_eval.py:176-242 (same signature): consensus always; segments and meta unless `concise`."""
    out = OutputTree(global_setting.output, suffix, concise, rna=getattr(global_setting, "mode", "dna") == "rna")
    written = out.consensus(file_pre, consensus, q_score)
    if not concise:
        out.segments(file_pre, segments, seg_q_score)
        out.meta(file_pre, len(written), time_list, global_setting)


# ------------------------------------------------------------------------------------------------
# cross-read batch packing: the feed side (chiron_eval.py:304-368)
# ------------------------------------------------------------------------------------------------
class Batch(object):
    """One packed batch.  `runs` lists its contiguous per-read chunks as (file name, first row, rows, index of the chunk's first
    window within its file); `fname` / `index` are the per-row tags of the reference (chiron_eval.py:328-329), expanded on
    demand: the pipeline itself only ever walks the runs (a per-row object array of names cost the main thread a quarter of
    its time per batch)."""
    __slots__ = ("_x", "pieces", "seq_len", "n_valid", "runs", "_rows")

    def __init__(self, x, seq_len, runs, n_valid, rows=None, pieces=None):
        """x: the [batch, segment_len] array, or None when `pieces` (the per-run arrays whose concatenation it is) is given: the
        engine takes the pieces as they are (Engine.submit_pieces) and nobody pays for the concatenation; `x` builds it on demand."""
        self._x, self.pieces, self.seq_len, self.runs, self.n_valid = x, pieces, seq_len, runs, n_valid
        self._rows = (len(x) if x is not None else sum(len(p) for p in pieces)) if rows is None else rows

    @property
    def x(self):
        if self._x is None:
            self._x = np.ascontiguousarray(np.concatenate(self.pieces, axis=0), dtype=np.float32)
        return self._x

    @classmethod
    def from_tags(cls, x, seq_len, fname, index, n_valid):
        """from the reference's per-row tags: rows with one name and one first-window index in a row are one run"""
        runs, pos, B = [], 0, len(fname)
        while pos < B:
            end = pos
            while end < B and fname[end] == fname[pos] and index[end] == index[pos]:
                end += 1
            if fname[pos] != "":
                runs.append((fname[pos], pos, end - pos, int(index[pos])))
            pos = end
        return cls(x, seq_len, runs, n_valid, rows=B)

    @property
    def fname(self):
        out = np.full(self._rows, "", dtype=object)
        for name, start, n, _ in self.runs:
            out[start:start + n] = name
        return out

    @property
    def index(self):
        out = np.full(self._rows, -1, dtype=np.int64)
        for _, start, n, first in self.runs:
            out[start:start + n] = first
        return out


class BatchPacker(object):
    """Append the windows of successive reads until exactly `batch_size` rows are present
    (chiron_eval.py:321-334).  Each row carries (file name, index of the chunk's first window within
    its file) exactly as the reference tags them (:328-329).  The final partial batch is padded
    by np.pad(mode='wrap') with tags -1 / '' (:352-360)."""

    def __init__(self, batch_size, segment_len, ratio):
        self.batch_size, self.segment_len, self.ratio = batch_size, segment_len, ratio
        self._reset()

    def _reset(self):
        self.x, self.sl, self.runs = [], [], []
        self.n = 0

    def add_read(self, name, event, event_length):
        """yield full batches while consuming one read's windows"""
        ds = signal_io.DataSet(event, event_length)
        i = 0
        if ds.reads_n == 0:
            return
        while ds.epochs_completed == 0:
            cur, cur_len, _ = ds.next_batch(self.batch_size - self.n, shuffle=False)
            n = len(cur)
            self.x.append(cur)
            self.sl.append(cur_len)
            self.runs.append((name, self.n, n, i))
            self.n += n
            i += n
            if self.n < self.batch_size:
                continue
            yield self._emit(self.n)

    def _emit(self, n_valid):
        sl = np.concatenate(self.sl, axis=0)
        if n_valid == self.batch_size and all(piece_ok(p, self.segment_len) for p in self.x):
            # a full batch: hand the per-run arrays on as they are (row slices of the reads' window arrays)
            b = Batch(None, seq_len_for_engine(sl, self.ratio), self.runs, n_valid, rows=n_valid, pieces=self.x)
            self._reset()
            return b
        x = np.concatenate(self.x, axis=0)
        if n_valid < self.batch_size:
            pad = self.batch_size - n_valid
            x = np.pad(x, ((0, pad), (0, 0)), mode="wrap")
            sl = np.pad(sl, (0, pad), mode="wrap")      # rows past n_valid carry no run: tags "" / -1
        b = Batch(np.ascontiguousarray(x, dtype=np.float32), seq_len_for_engine(sl, self.ratio), self.runs, n_valid)
        self._reset()
        return b

    def flush(self):
        if self.n > 0:
            return self._emit(self.n)
        return None


class ReadCollector(object):
    """The drain side (chiron_eval.py:403-446): split each decoded batch into contiguous per-file runs,
    re-index rows, and release a read once all of its windows have arrived, pieces ordered by the
    within-file index of their first window (intended semantic)."""

    def __init__(self):
        self.val = {}

    def expect(self, name, reads_n, meta):
        self.val.setdefault(name, {"total": 0, "pieces": {}})
        self.val[name]["reads_n"] = reads_n
        self.val[name]["meta"] = meta

    def add_batch(self, batch, result, want_qs):
        """-> list of (name, reads [ragged int arrays], qs_list [n,1], meta) for completed reads"""
        done = []
        compact = getattr(result, "compact", None)
        if compact is not None:
            return self._add_batch_compact(batch, compact, result.prob_logits, want_qs)
        predict_val = ([result.decoded], result.log_prob)
        runs = batch.runs
        k = 0
        while k < len(runs):
            fn, pos, n, first_idx = runs[k]
            end = pos + n
            k += 1
            while k < len(runs) and runs[k][0] == fn and runs[k][1] == end:   # adjacent rows with one name are one run (:415-436)
                end += runs[k][2]
                k += 1
            if fn != "":
                sliced = slice_ctc_decoding_result(predict_val, pos, end)
                rec = self.val.setdefault(fn, {"total": 0, "pieces": {}})
                rec["pieces"][int(first_idx)] = (sliced, result.prob_logits[pos:end])
                rec["total"] += end - pos
                if "reads_n" in rec and rec["total"] == rec["reads_n"]:
                    done.append(self._finish(fn, want_qs))
        return done

    def _add_batch_compact(self, batch, compact, prob_logits, want_qs):
        """add_batch on the engine's compact decode (Engine.collect().compact: the rows' labels back to back + labels per row):
        a run of rows is one slice of `flat`, its windows' lengths are the non-zero counts, the rows that decoded to something
        are where the counts are non-zero -- the same pieces slice_ctc_decoding_result + _rows_of extract from the SparseTensor
        (chiron_eval.py:36-98), without scanning 16 bytes of (row, position) per base."""
        done = []
        counts = compact.counts
        off = np.zeros(counts.shape[0] + 1, dtype=np.int64)
        np.cumsum(counts, out=off[1:])
        runs = batch.runs
        k = 0
        while k < len(runs):
            fn, pos, n, first_idx = runs[k]
            end = pos + n
            k += 1
            while k < len(runs) and runs[k][0] == fn and runs[k][1] == end:
                end += runs[k][2]
                k += 1
            if fn != "":
                c = counts[pos:end]
                nz = c > 0
                rec = self.val.setdefault(fn, {"total": 0, "pieces": {}})
                rec["pieces"][int(first_idx)] = (compact.flat[off[pos]:off[end]], c[nz].astype(np.int64), prob_logits[pos:end][nz] if want_qs else None)
                rec["total"] += end - pos
                if "reads_n" in rec and rec["total"] == rec["reads_n"]:
                    done.append(self._finish(fn, want_qs))
        return done

    def _finish(self, name, want_qs):
        """-> (name, flat, seg_len, qs_list, meta): the read's decoded windows as ONE uint8 array of base indices and the
        length of every window that decoded to something (rows with an empty decode vanish, as in sparse2dense,
        chiron_eval.py:36-66); qs_list holds the path_prob of exactly those rows."""
        rec = self.val.pop(name)
        vals, lens, qss = [], [], []
        for i in sorted(rec["pieces"]):
            if len(rec["pieces"][i]) == 3:          # a compact piece: (flat labels, lengths of the non-empty windows, their path_prob)
                flat_i, lens_i, qs_i = rec["pieces"][i]
                vals.append(flat_i)
                lens.append(lens_i)
                if want_qs:
                    qss.append(qs_i)
                continue
            (decoded, _), logits_prob = rec["pieces"][i]
            ids, starts, ends = _rows_of(decoded[0])
            vals.append(decoded[0].values)
            lens.append(ends - starts)
            if want_qs:
                qss.append(logits_prob[ids])
        flat = np.concatenate(vals).astype(np.uint8) if vals else np.zeros(0, dtype=np.uint8)
        seg_len = np.concatenate(lens).astype(np.int64) if lens else np.zeros(0, dtype=np.int64)
        qs_list = np.concatenate(qss) if qss else np.empty((0, 1), dtype=float)
        return name, flat, seg_len, qs_list, rec["meta"]


def list_inputs(input_path, recursive):
    """chiron_eval.py:277-293, sorted (Q6)."""
    if os.path.isdir(input_path):
        files = []
        if recursive:
            for dirpath, _, filenames in os.walk(input_path):
                for fn in filenames:
                    files.append(os.path.relpath(os.path.join(dirpath, fn), input_path))
        else:
            files = os.listdir(input_path)
        return sorted(files), input_path
    return [os.path.basename(input_path)], os.path.abspath(os.path.join(input_path, os.path.pardir))


def finish_read(name, reads, qs_list, FLAGS, t_start, reading_time):
    """chiron_eval.py:446-462: bases, consensus vote, quality string, writers."""
    file_pre = os.path.splitext(name)[0]
    basecall_time = time.time() - t_start
    bpreads = bases_of_reads(reads)
    js_ratio = FLAGS.jump / FLAGS.segment_len
    kernal = get_assembler_kernal(FLAGS.jump, FLAGS.segment_len)
    qs_string = None
    if kernal != "simple" and len(bpreads) >= getattr(FLAGS, "device_vote_min_segments", DEVICE_VOTE_MIN_SEGMENTS):
        # very long reads: displacements, vote, argmax and quality string in one pass on the GPU
        c_bpread, qs_string = assembly.consensus_device(bpreads, qs_list if FLAGS.extension == "fastq" else None, kernal,
                                                        getattr(FLAGS, "device", 0))
    elif FLAGS.extension == "fastq":
        consensus, qs_consensus = assembly.simple_assembly_qs(bpreads, qs_list, js_ratio, kernal=kernal)
        qs_string = qs(consensus, qs_consensus)
        c_bpread = index2base(np.argmax(consensus, axis=0))
    else:
        consensus = assembly.simple_assembly(bpreads, js_ratio, kernal=kernal)
        c_bpread = index2base(np.argmax(consensus, axis=0))
    assembly_time = time.time() - t_start
    write_output(bpreads, c_bpread, [t_start, reading_time, basecall_time, assembly_time], file_pre,
                 concise=FLAGS.concise, suffix=FLAGS.extension, q_score=qs_string, global_setting=FLAGS)
    return c_bpread


def split_flat(flat, seg_len):
    """the ragged list of windows behind the flat form (one array of base indices per window)"""
    ends = np.cumsum(seg_len)
    return [flat[e - n:e] for e, n in zip(ends.tolist(), np.asarray(seg_len).tolist())]


def finish_read_flat(name, flat, seg_len, qs_list, FLAGS, t_start, reading_time):
    """finish_read on the flat form of a read's decoded windows (all bases as one uint8 array + the length of every
    window): one call into chiron_finish_read (bases, vote, argmax, quality string, result/ and segments/ files, all
    without the interpreter and without the GIL), then the meta file.  Same files as finish_read, byte for byte (tests).
    Reads long enough for the device vote, and anything the native writer does not cover, take finish_read."""
    import ctypes as C
    from . import _lib
    kernal = get_assembler_kernal(FLAGS.jump, FLAGS.segment_len)
    n_seg = int(seg_len.shape[0])
    if getattr(FLAGS, "python_finish", False) or \
            (kernal != "simple" and n_seg >= getattr(FLAGS, "device_vote_min_segments", DEVICE_VOTE_MIN_SEGMENTS)):
        return finish_read(name, split_flat(flat, seg_len), qs_list, FLAGS, t_start, reading_time)
    file_pre = os.path.splitext(name)[0]
    basecall_time = time.time() - t_start
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    off = np.zeros(n_seg + 1, dtype=np.int64)
    np.cumsum(seg_len, out=off[1:])
    fastq = FLAGS.extension == "fastq"
    qs_arr = np.ascontiguousarray(np.asarray(qs_list, dtype=np.float64).reshape(n_seg, -1)[:, 0]) if (fastq and n_seg) else None
    if fastq and qs_arr is None:
        qs_arr = np.zeros(1, dtype=np.float64)
    cons = np.empty(max(int(flat.shape[0]), 1), dtype=np.uint8)
    n = C.c_int64()
    root, ext = FLAGS.output, FLAGS.extension
    args = lambda: (flat.ctypes.data, off.ctypes.data, n_seg, None if qs_arr is None else qs_arr.ctypes.data, assembly._kernal_id(kernal), 0.2,
                    FLAGS.jump / FLAGS.segment_len, file_pre.encode(), os.path.join(root, "result", file_pre + "." + ext).encode(),
                    None if FLAGS.concise else os.path.join(root, "segments", file_pre + "." + ext).encode(), 1 if fastq else 0,
                    1 if getattr(FLAGS, "mode", "dna") == "rna" else 0, cons.ctypes.data, cons.shape[0], C.byref(n))
    lib = _lib.load()
    st = lib.chiron_finish_read(*args())
    if st != _lib.OK:
        # a recursive input puts sub-folders into file_pre: create them on demand and retry once
        for sub in ("result",) + (() if FLAGS.concise else ("segments", "meta")):
            os.makedirs(os.path.dirname(os.path.join(root, sub, file_pre)), exist_ok=True)
        _lib.check(lib.chiron_finish_read(*args()))
    assembly_time = time.time() - t_start
    if not FLAGS.concise:
        OutputTree(root, ext, False).meta(file_pre, n.value, (t_start, reading_time, basecall_time, assembly_time), FLAGS)
    return cons[:n.value].tobytes().decode("ascii")


_FLAG_FIELDS = ("input", "output", "model", "start", "batch_size", "segment_len", "jump", "extension", "concise", "mode", "device",
                "device_vote_min_segments")


def _finish_read_in_process(name, flat, seg_len, qs_list, flags, t_start, reading_time):
    """finish_read for a worker PROCESS (FLAGS.finish_procs > 0): the decoded windows travel as one flat base array +
    lengths, the settings as a plain dict.  The worker never touches the GPU."""
    import argparse
    flags = dict(flags, device_vote_min_segments=1 << 62)     # host vote only: a worker process must not open the GPU
    return finish_read_flat(name, flat, seg_len, qs_list, argparse.Namespace(**flags), t_start, reading_time)


def _record_engine(FLAGS, engine):
    """<output>/log/engine[.rank<r>].json: what basecalled this output -- dtype, and whether the fp16 bias correction was applied (the
    same read decodes differently with and without it; the reference's meta/ files keep the reference's format)."""
    import json
    try:
        d = os.path.join(FLAGS.output, "log")
        os.makedirs(d, exist_ok=True)
        rank = os.environ.get("CHIRON_LOCAL_RANK", os.environ.get("RANK"))
        with open(os.path.join(d, "engine%s.json" % (".rank" + rank if rank else "")), "w") as fh:
            json.dump({"dtype": engine.dtype, "fp16_bias_correction": bool(engine.calibrated), "max_batch": engine.max_batch,
                       "segment_len": engine.segment_len, "slots": engine.n_slots, "device": engine.device_id}, fh)
    except OSError:
        pass


def native_pipeline_ok(FLAGS, engine, fast5_files, signal_names=None):
    """Whether this call can run on chiron_pipeline_run (csrc/pipeline.cpp: the same pipeline with reader / packer / finisher in C++
    threads, no interpreter lock): the direct fast5 path (raw DAC counts, names from the file stems) or a folder of `.signal` text files,
    a real Engine (or the null engine of the host-ceiling measurement) with population BN, finishers in threads, host vote.  FLAGS.python_pipeline = True (or
    CHIRON_PYTHON_PIPELINE=1) keeps the Python pools -- the reference implementation the native one is tested against."""
    if getattr(FLAGS, "python_pipeline", False) or os.environ.get("CHIRON_PYTHON_PIPELINE") == "1":
        return False
    if fast5_files is None:          # a folder of `.signal` files (what extraction leaves under raw/, or the caller's own): all of them text
        if not signal_names or not all(n.endswith(".signal") for n in signal_names) or getattr(FLAGS, "reverse_fast5", False):
            return False
    elif getattr(FLAGS, "unit", False) or getattr(FLAGS, "idname", False):      # extraction options the native reader does not cover
        return False
    if getattr(FLAGS, "python_finish", False) or int(getattr(FLAGS, "finish_procs", 0) or 0) > 0:
        return False
    if getattr(engine, "null_engine", False):
        return True
    return isinstance(engine, Engine) and engine.spec.bn_mode == "population"


class NativeResults(dict):
    """What evaluation() returns after a native run: read name (+ ".signal", the Python pipeline's keys) -> consensus string, like the
    dict the Python pipeline builds -- but the strings stay in result/<name>.<ext> until somebody asks for one (a sharded `chiron call`
    never does: 10 000 reads are 10 000 small files nobody needs to read back)."""

    def __init__(self, names, result_dir, ext, rna):
        dict.__init__(self, ((n + ".signal", None) for n in names))
        self._dir, self._ext, self._rna = result_dir, ext, rna

    def _load(self, key):
        text = open(os.path.join(self._dir, key[:-len(".signal")] + self._ext)).read().split("\n")
        seq = text[1] if len(text) > 1 else ""
        return seq.replace("U", "T").replace("u", "t") if self._rna else seq      # the returned string is the consensus before T -> U (finish_read)

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None:
            v = self._load(key)
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __eq__(self, other):
        return dict(self.items()) == (dict(other.items()) if isinstance(other, dict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


def run_native_pipeline(FLAGS, engine, fast5_files, n_threads, name_root=None):
    """chiron_pipeline_run behind evaluation(): -> NativeResults (read name + ".signal" -> consensus string, read from result/ on
    demand).  Skipped files are logged the way extract.extract_records logs them."""
    import ctypes as C
    from . import _lib
    from . import extract as extract_mod
    lib = _lib.load()
    null = bool(getattr(engine, "null_engine", False))
    paths = (C.c_char_p * len(fast5_files))(*[os.fsencode(p) for p in fast5_files])
    opts = _lib.PipelineOpts(FLAGS.batch_size, FLAGS.segment_len, FLAGS.jump, FLAGS.start, FLAGS.beam, int(FLAGS.extension == "fastq"),
                             int(bool(FLAGS.concise)), int(getattr(FLAGS, "mode", "dna") == "rna"), int(bool(getattr(FLAGS, "no_raw", False))),
                             n_threads, engine.n_slots, int(null), float(engine.ratio), os.fsencode(FLAGS.output),
                             getattr(FLAGS, "delimiter", "\n").encode(), str(FLAGS.input).encode(), str(FLAGS.model).encode(),
                             None if name_root is None else os.fsencode(name_root))
    stats = _lib.PipelineStats()
    st = lib.chiron_pipeline_run(None if null else engine._h, paths, len(fast5_files), C.byref(opts), C.byref(stats))
    for line in stats.messages.decode("utf-8", "replace").splitlines():
        extract_mod.logger.error(line)
    if st != _lib.OK:
        raise _lib.ChironError(st, stats.messages.decode("utf-8", "replace") or lib.chiron_last_error().decode("utf-8", "replace"))
    res_dir, ext = os.path.join(FLAGS.output, "result"), "." + FLAGS.extension
    names = sorted(os.path.relpath(os.path.join(dp, n), res_dir)[:-len(ext)] for dp, _, fns in os.walk(res_dir) for n in fns if n.endswith(ext))
    evaluation.last_native_stats = {k: getattr(stats, k) for k in ("reads", "reads_finished", "windows", "batches", "consensus_bases", "files_failed", "seconds")}
    return NativeResults(names, res_dir, ext, getattr(FLAGS, "mode", "dna") == "rna")


def evaluation(FLAGS, engine=None, file_list=None, fast5_files=None):
    """chiron_eval.py:378-463 on one GPU.  `file_list` restricts the reads this process handles
    (per-read sharding across GPUs).  `fast5_files` (full paths) switches to the direct fast5 path of
    `chiron call`: every file is decoded by the native reader in a reader thread, its raw/<name>.signal
    and reference files are written as the reference's extraction writes them (extract.extract_records), and the decoded
    samples are windowed straight away -- the text is never parsed back.  Output files are the same either way."""
    own_engine = engine is None
    if own_engine:
        spec, weights, _ = model_mod.load_model(FLAGS.model, allow_synthetic=getattr(FLAGS, "synthetic_weights", False))
        engine = Engine(spec, weights, max_batch=FLAGS.batch_size, segment_len=FLAGS.segment_len,
                        device_id=getattr(FLAGS, "device", 0), n_slots=int(getattr(FLAGS, "slots", 0) or 3), max_beam=FLAGS.beam,
                        dtype=getattr(FLAGS, "dtype", "fp32"),
                        # fp16: bias correction for the weights' rounding to halves on a FIXED synthetic calibration batch, so every
                        # rank of a sharded run builds the same engine and a read's output does not depend on which process
                        # basecalls it (Engine(calibrate=...) is the one switch of every entry point)
                        calibrate=not getattr(FLAGS, "no_calibration", False))
        _record_engine(FLAGS, engine)
    if fast5_files is None:
        files, file_dir = list_inputs(FLAGS.input, getattr(FLAGS, "recursive", False))
        if file_list is not None:
            files = [f for f in files if f in set(file_list)]
    else:
        files, file_dir = [], None
    for sub in ("segments", "result", "meta"):
        os.makedirs(os.path.join(FLAGS.output, sub), exist_ok=True)
    want_qs = FLAGS.extension == "fastq"
    packer = BatchPacker(FLAGS.batch_size, FLAGS.segment_len, engine.ratio)
    collector = ReadCollector()
    inflight = [None] * engine.n_slots
    results = {}
    step = [0]
    # The reference overlaps reading, inference and decoding with TF queue runners (chiron_eval.py:304-368, :465-522).
    # Here: a pool of reader threads parses / windows the next files while the engine works (the native text parser
    # and zlib release the GIL), and finished reads are assembled and written on a second pool; the main thread only
    # packs batches and talks to the engine.  Batches are packed in file order, so results do not depend on timing.
    n_threads = int(getattr(FLAGS, "threads", 0) or 0)
    if n_threads <= 0:
        # default: a quarter of this rank's share of the host's cores, between 4 and 6.  Round 5 measured, on a 2 x 64-core host: behind
        # the null engine one rank moves 540 / 800 / 700 / 670 / 560 k windows/s with 2 / 4 / 8 / 12 / 24 reader + finisher threads (more
        # threads = more interpreter-lock hand-overs, not more work done: the native calls are short), eight ranks 2.6 / 2.3 / 2.2 M with
        # 4 / 8 / 16; behind the real fp16 engine at batch 4096 with raw/*.signal written: 4 threads 17 .. 19 Mbases/s (too few while a
        # thread sits in a 0.4 MB text write), 6 threads 24.2 / 24.5, 12 threads 24.3 / 13.3 (unstable).  -t overrides.
        ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
        n_threads = min(6, max(4, (os.cpu_count() or 16) // (4 * ranks_here)))
    signal_names = None if fast5_files is not None else [n for n in files if n.endswith(".signal") or n.endswith(".fast5")]
    if native_pipeline_ok(FLAGS, engine, fast5_files, signal_names):
        # the whole host side in one native call: C++ reader / finisher threads, this thread packs and talks to the engine (csrc/pipeline.cpp)
        if int(getattr(FLAGS, "threads", 0) or 0) <= 0:
            # -t 0: C++ threads scale with their number on a rank of its own (2 / 3 / 6 / 12 reader + finisher threads: 0.61 / 0.92 / 1.80 /
            # 3.35 M windows/s behind the null engine) -- and EIGHT ranks writing three files + 0.4 MB of raw/ text per read into one file
            # system are fastest with few (1 / 2 / 3 / 4 / 6 / 10 per rank: 2.47 / 4.31 / 4.40 / 3.77 / 3.53 / 3.24 M in total): about 24
            # reader threads per host (profiles/r06_host_ceiling.json)
            n_threads = max(2, min(12, 24 // ranks_here, max(2, (os.cpu_count() or 8) // 2)))
        try:
            if fast5_files is not None:
                return run_native_pipeline(FLAGS, engine, list(fast5_files), n_threads)
            return run_native_pipeline(FLAGS, engine, [os.path.join(file_dir, n) for n in signal_names], n_threads, name_root=file_dir)
        finally:
            if own_engine:
                engine.close()
    readers = ThreadPoolExecutor(max_workers=n_threads)
    # Finishing (base strings, consensus vote, quality string, three files per read) is Python + numpy + native calls: as
    # threads it is bound by the GIL at a few hundred reads per second, enough for the fp32 engine.  FLAGS.finish_procs > 0
    # moves it to that many worker processes (spawned: they import the host modules only, never the engine).
    n_procs = int(getattr(FLAGS, "finish_procs", 0) or 0)
    if n_procs > 0:
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        finishers = ProcessPoolExecutor(max_workers=n_procs, mp_context=multiprocessing.get_context("spawn"))
        flag_dict = {k: getattr(FLAGS, k) for k in _FLAG_FIELDS if hasattr(FLAGS, k)}
    else:
        finishers = ThreadPoolExecutor(max_workers=n_threads)
    finishing = []

    # the decode in the regroup's own form (rows' labels back to back + labels per row) where the engine offers it
    import inspect
    compact_kw = {"compact": True} if "compact" in inspect.signature(engine.submit).parameters else {}

    def collect(slot):
        """-> (batch, result) of the slot's finished batch, or None: the slot is free for the next submit afterwards"""
        if inflight[slot] is None:
            return None
        batch, inflight[slot] = inflight[slot], None
        return batch, engine.collect(slot)

    def regroup(done):
        if done is None:
            return
        batch, res = done
        for name, flat, seg_len, qs_list, meta in collector.add_batch(batch, res, want_qs):
            if n_procs > 0:
                fut = finishers.submit(_finish_read_in_process, name, flat, seg_len, qs_list, flag_dict, meta[0], meta[1])
            else:
                fut = finishers.submit(finish_read_flat, name, flat, seg_len, qs_list, FLAGS, meta[0], meta[1])
            finishing.append((name, fut))

    def launch(batch):
        # the slot gets its next batch before the host regroups the finished one: the stream never waits for Python
        slot = step[0] % engine.n_slots
        step[0] += 1
        done = collect(slot)
        if batch.pieces is not None and hasattr(engine, "submit_pieces"):
            engine.submit_pieces(slot, batch.pieces, batch.seq_len, beam_width=FLAGS.beam, want_prob=want_qs, compact=True)
        else:
            engine.submit(slot, batch.x, batch.seq_len, beam_width=FLAGS.beam, want_prob=want_qs, **compact_kw)
        inflight[slot] = batch
        regroup(done)

    def load(name):
        """one input file -> [(read name, DataSet, start time, reading time)]"""
        t0 = time.time()
        if fast5_files is not None:
            from . import extract as extract_mod
            out = []
            for rname, sig in extract_mod.extract_records(name, FLAGS):       # writes raw/<rname>.signal as extraction does
                ev, ln = signal_io.window_signal(sig, FLAGS.start, FLAGS.jump, FLAGS.segment_len)
                out.append((rname + ".signal", signal_io.DataSet(ev, ln), t0, time.time() - t0))
                t0 = time.time()
            return out
        ds = signal_io.read_data_for_eval(os.path.join(file_dir, name), FLAGS.start, seg_length=FLAGS.segment_len,
                                          step=FLAGS.jump, reverse_fast5=getattr(FLAGS, "reverse_fast5", False))
        return [(name, ds, t0, time.time() - t0)]

    names = list(fast5_files) if fast5_files is not None else [n for n in files if n.endswith(".signal") or n.endswith(".fast5")]
    ahead = collections.deque()
    nxt = 0
    try:
        for _ in names:
            while nxt < len(names) and len(ahead) < 2 * n_threads:
                ahead.append(readers.submit(load, names[nxt]))
                nxt += 1
            for name, ds, t0, t_read in ahead.popleft().result():
                collector.expect(name, ds.reads_n, (t0, t_read))
                if ds.reads_n == 0:
                    results[name] = finish_read(name, [], np.empty((0, 1)), FLAGS, t0, t_read)     # (the Python writers)
                    collector.val.pop(name, None)
                    continue
                for batch in packer.add_read(name, ds.event, ds.event_length):
                    launch(batch)
        last = packer.flush()
        if last is not None:
            launch(last)
        for slot in range(engine.n_slots):
            regroup(collect((step[0] + slot) % engine.n_slots))
        for name, fut in finishing:
            results[name] = fut.result()
    finally:
        readers.shutdown(wait=True)
        finishers.shutdown(wait=True)
    if own_engine:
        engine.close()
    return results


def run(args):
    """chiron_eval.py:525-544."""
    FLAGS = args
    print("The result will be written to %s" % (FLAGS.output))
    os.makedirs(FLAGS.output, exist_ok=True)
    time_dict = unix_time(evaluation, FLAGS, fast5_files=getattr(FLAGS, "fast5_files", None))
    print("Real time:%5.3f Systime:%5.3f Usertime:%5.3f" % (time_dict["real"], time_dict["sys"], time_dict["user"]))
    meta_folder = os.path.join(FLAGS.output, "meta")
    os.makedirs(meta_folder, exist_ok=True)
    file_pre = "all" if os.path.isdir(FLAGS.input) else os.path.splitext(os.path.basename(FLAGS.input))[0]
    with open(os.path.join(meta_folder, file_pre + ".meta"), "a+") as out_meta:
        out_meta.write("# Wall_time Sys_time User_time Cpu_time\n")
        out_meta.write("%5.3f %5.3f %5.3f %5.3f\n" % (time_dict["real"], time_dict["sys"], time_dict["user"],
                                                      time_dict["sys"] + time_dict["user"]))
    return time_dict
