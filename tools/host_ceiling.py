#!/usr/bin/env python3
"""The host pipeline of `chiron call` WITHOUT a GPU: how many windows per second can one rank's host side move, and how does
that scale when 1 / 2 / 4 / 8 ranks share one host (BASELINE configs[3]: 8 ranks per node)?

Everything of `chiron_amd.eval.evaluation` runs as in production -- reader threads (native fast5 decode + raw/<name>.signal copy,
or the .signal text parser), windowing, cross-read batch packing, the per-read regroup of the decoded SparseTensor, the native
finisher (vote, quality string, result / segments / meta files) -- only the engine is a NULL engine: submit() does nothing,
collect() returns a canned decode of the batch (a fresh copy of a SparseTensor with ~44 bases per window, what a trained model
emits at 8.9 samples per base; path_prob per window) after an optional sleep (--engine-ms, default 0: the host's ceiling).
So the figure is the rate at which the host can feed and drain an infinitely fast GPU: it has to stay above 8 x 104 k windows/s
(fp32 engine) and ideally 8 x 550 k (fp16 engine, batch 4096) on the node that runs eight ranks.

Ranks are separate processes, each pinned to its slice of the host's cores exactly as `chiron call --gpus N` pins its ranks
(chiron_amd.shard.rank_cpus), each with its own reads and its own output folder (tmpfs if --workdir points there).

    python tools/host_ceiling.py [--reads 256] [--ranks 1,2,4,8] [--inputs fast5,signal] [--batch 1100] [--engine-ms 0]
        -> one JSON line per (input, ranks): windows/s per rank and in total, plus gpurun_out/host_ceiling.json
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BASES_PER_WINDOW = 44


class NullEngine(object):
    """The surface of chiron_amd.Engine that eval.evaluation uses, with no device behind it.  null_engine = True (the default since round
    6) makes evaluation() run the fast5 path on chiron_pipeline_run (csrc/pipeline.cpp: C++ reader / packer / finisher threads) with ITS
    null engine -- the same canned ~44 bases per window; HOST_CEILING_PYTHON=1 keeps the Python thread pools (rounds 4 / 5)."""
    null_engine = os.environ.get("HOST_CEILING_PYTHON") != "1"

    def __init__(self, max_batch, segment_len, n_slots=3, engine_ms=0.0, seed=5):
        from chiron_amd.engine import SparseTensor, DecodeResult
        self._ST, self._DR = SparseTensor, DecodeResult
        self.ratio, self.T, self.n_slots, self.max_batch, self.segment_len = 1.0, segment_len, n_slots, max_batch, segment_len
        self.engine_ms = engine_ms
        rng = np.random.RandomState(seed)
        n = rng.randint(BASES_PER_WINDOW - 6, BASES_PER_WINDOW + 7, size=max_batch)
        rows = np.repeat(np.arange(max_batch, dtype=np.int64), n)
        pos = np.concatenate([np.arange(k, dtype=np.int64) for k in n])
        self._idx = np.stack([rows, pos], axis=1)
        self._val = rng.randint(0, 4, size=rows.shape[0]).astype(np.int64)
        self._ends = np.cumsum(n)
        self._prob = rng.uniform(1.0, 6.0, size=(max_batch, 1)).astype(np.float32)
        self._lp = np.zeros((max_batch, 1), dtype=np.float32)
        self._shape_cols = int(n.max())
        self._val8, self._n32 = self._val.astype(np.uint8), n.astype(np.int32)
        self._stage = np.empty((max_batch, segment_len), dtype=np.float32)
        self.legacy = os.environ.get("HOST_CEILING_LEGACY") == "1"     # round 4's host path: concatenated batches, SparseTensor regroup
        self._pending = [None] * n_slots
        if not self.legacy:
            self.submit_pieces = self._submit_pieces

    def submit(self, slot, x, seq_len, beam_width=0, want_prob=True, want_logits=False, copy_decoded=True, compact=False):
        self._pending[slot] = (int(x.shape[0]), time.time(), compact and not self.legacy)

    def _submit_pieces(self, slot, pieces, seq_len, beam_width=0, want_prob=True, compact=True):
        # what chiron_engine_submit_pieces does on the host: the pieces into the slot's staging buffer
        row = 0
        for p in pieces:
            self._stage[row:row + len(p)] = p
            row += len(p)
        self._pending[slot] = (row, time.time(), compact)

    def collect(self, slot):
        b, t0, compact = self._pending[slot]
        self._pending[slot] = None
        if self.engine_ms > 0:
            left = self.engine_ms * 1e-3 - (time.time() - t0)
            if left > 0:
                time.sleep(left)
        nnz = int(self._ends[b - 1])
        if compact:
            from chiron_amd.engine import CompactDecode
            return self._DR(None, self._lp[:b].copy(), self._prob[:b].copy(), None,
                            CompactDecode(self._val8[:nnz].copy(), self._n32[:b].copy(), np.asarray([b, self._shape_cols], dtype=np.int64)))
        # fresh arrays per batch, as Engine.collect's copies out of the slot's pinned buffers are
        return self._DR(self._ST(self._idx[:nnz].copy(), self._val[:nnz].copy(), np.asarray([b, self._shape_cols], dtype=np.int64)),
                        self._lp[:b].copy(), self._prob[:b].copy(), None)

    def close(self):
        pass


def write_inputs(folder, n_reads, kind, seed):
    import chiron_amd as ca
    from chiron_amd import fast5
    os.makedirs(folder, exist_ok=True)
    sig = ca.synthetic_signal(n_reads, 100000, seed=seed)
    if kind == "fast5":
        import h5_writer
        for i in range(n_reads):
            h5_writer.write_multi_read_fast5(os.path.join(folder, "read%05d.fast5" % i), [("", "id-%d" % i, sig[i].astype(np.int16), None)], chunk=20000)
    else:
        for i in range(n_reads):
            fast5.write_signal_text(os.path.join(folder, "read%05d.signal" % i), sig[i], "\n")


def rank_main(a):
    """one rank: pin, wait for the start file, run the pipeline once untimed on a few reads (page cache, imports), then timed"""
    from chiron_amd import shard, eval as ce, extract as ex
    try:
        os.sched_setaffinity(0, shard.rank_affinity(a.rank, a.world) if os.environ.get("HOST_CEILING_PLAIN_SLICES") != "1" else shard.rank_cpus(a.rank, a.world))
    except (AttributeError, OSError):
        pass
    os.environ["LOCAL_WORLD_SIZE"] = str(a.world)            # the default reader / finisher thread count follows it (eval.evaluation)

    class F(object):
        start, segment_len, jump, batch_size = 0, 400, 390, a.batch
        extension, concise, mode, recursive = "fastq", os.environ.get("HOST_CEILING_CONCISE") == "1", "dna", True
        unit, idname, delimiter, test_number = False, False, "\n", None
        beam, threads, finish_procs, model = 0, a.threads, 0, "null-engine"
        no_raw = os.environ.get("HOST_CEILING_NO_RAW") == "1"
    F.input = F.input_dir = a.input
    F.output = F.output_dir = a.output

    def once(files_limit=None):
        shutil.rmtree(a.output, ignore_errors=True)
        eng = NullEngine(a.batch, 400, 3, a.engine_ms)
        if a.kind == "fast5":
            ex.prepare_folders(F)
            files = ex.list_fast5(a.input)[:files_limit]
            return len(ce.evaluation(F, engine=eng, fast5_files=files))
        names = sorted(os.listdir(a.input))[:files_limit]
        return len(ce.evaluation(F, engine=eng, file_list=names))

    once(8)
    open(os.path.join(a.sync, "ready.%d" % a.rank), "w").close()
    while not os.path.exists(os.path.join(a.sync, "go")):
        time.sleep(0.002)
    t0 = time.time()
    n = once()
    dt = time.time() - t0
    print(json.dumps({"rank": a.rank, "reads": n, "seconds": dt, "t_start": t0, "t_end": t0 + dt}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=256, help="reads of 100k samples (257 windows) per rank")
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--inputs", default="fast5,signal")
    ap.add_argument("--batch", type=int, default=1100)
    ap.add_argument("--engine-ms", type=float, default=0.0, help="time a batch spends in the null engine (0: the host's ceiling)")
    ap.add_argument("--threads", type=int, default=0, help="reader / finisher threads per rank (0: the product's default)")
    ap.add_argument("--workdir", default=None)
    # internal: one rank
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--kind", default="fast5")
    ap.add_argument("--input", default=None)
    ap.add_argument("--output", default=None)
    ap.add_argument("--sync", default=None)
    a = ap.parse_args()
    if a.rank >= 0:
        return rank_main(a)

    wd = a.workdir or tempfile.mkdtemp(prefix="host_ceiling_")
    results = []
    windows_per_read = -(-100000 // 390)
    try:
        for kind in a.inputs.split(","):
            max_ranks = max(int(r) for r in a.ranks.split(","))
            for r in range(max_ranks):
                write_inputs(os.path.join(wd, kind, "in%d" % r), a.reads, kind, seed=77 + r)
            for world in [int(r) for r in a.ranks.split(",")]:
                sync = os.path.join(wd, "sync_%s_%d" % (kind, world))
                shutil.rmtree(sync, ignore_errors=True)
                os.makedirs(sync)
                procs = []
                for r in range(world):
                    cmd = [sys.executable, os.path.abspath(__file__), "--rank", str(r), "--world", str(world), "--kind", kind, "--batch", str(a.batch),
                           "--engine-ms", str(a.engine_ms), "--threads", str(a.threads), "--input", os.path.join(wd, kind, "in%d" % r),
                           "--output", os.path.join(wd, kind, "out%d_%d" % (world, r)), "--sync", sync]
                    procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, universal_newlines=True))
                while len([n for n in os.listdir(sync) if n.startswith("ready.")]) < world:
                    if any(p.poll() not in (None, 0) for p in procs):
                        raise RuntimeError("a rank failed before the start")
                    time.sleep(0.01)
                open(os.path.join(sync, "go"), "w").close()
                recs = []
                for p in procs:
                    out, _ = p.communicate()
                    if p.returncode != 0:
                        raise RuntimeError("rank failed")
                    recs.append(json.loads(out.strip().split("\n")[-1]))
                wall = max(x["t_end"] for x in recs) - min(x["t_start"] for x in recs)
                total = sum(x["reads"] for x in recs) * windows_per_read
                rec = {"input": kind, "ranks": world, "reads_per_rank": a.reads, "batch": a.batch, "engine_ms": a.engine_ms,
                       "windows_per_s_total": round(total / wall), "windows_per_s_per_rank": round(total / wall / world),
                       "slowest_rank_s": round(max(x["seconds"] for x in recs), 3), "fastest_rank_s": round(min(x["seconds"] for x in recs), 3),
                       "host_cores": len(os.sched_getaffinity(0)), "needed_fp32_total": 104000 * world, "needed_fp16_total": 550000 * world}
                results.append(rec)
                print(json.dumps(rec))
                sys.stdout.flush()
                for r in range(world):
                    shutil.rmtree(os.path.join(wd, kind, "out%d_%d" % (world, r)), ignore_errors=True)
    finally:
        if a.workdir is None:
            shutil.rmtree(wd, ignore_errors=True)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(results, open(os.path.join(out_dir, "host_ceiling.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
