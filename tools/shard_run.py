#!/usr/bin/env python3
"""BASELINE configs[3] end to end: N synthetic reads x S samples -> `chiron call` on 1 rank and on R ranks
(one process per GPU under torch.distributed.run, reads sharded per rank, host-side gather) -> the merged FASTA/FASTQ
and every result file must be byte-identical (meta/ holds timings and is not compared).

    python tools/shard_run.py --reads 10000 --samples 100000 --ranks 8 --chunk 1000 --workdir /scratch/c3   # as written
    python tools/shard_run.py --reads 24 --ranks 2 --share-gpu                    # self-test on a 1-GPU box

Bounded and resumable: the reads are processed in chunks of --chunk reads (default: all at once).  A chunk's inputs are
written, basecalled on 1 rank and on R ranks, compared, recorded in <workdir>/state.json (timings, SHA-256 of the merged
file) and DELETED before the next chunk starts, so the disk holds one chunk at a time -- with --input fast5 (deflated
int16, ~0.13 MB per 100k-sample read) plus the two output trees (raw/*.signal 0.4 MB per read each; result / segments
~0.03 MB): about 1 GB per 1000 reads.  A re-run with the same --workdir skips the chunks state.json already holds.
Host load to expect on an 8-GPU node: every rank runs 4 reader threads (-t 4: native fast5 decode / text parse and the
windowing), 4 finishing threads (native vote + writers, GIL-free) and its main thread; 8 ranks = 72 threads, of which
about 2 cores per rank are busy at the fp32 engine's rate (94 k windows/s per GPU) -- 16 of the node's cores.

--share-gpu puts every rank on device 0 (CHIRON_SHARE_GPU=1: gloo carries the barriers, the engines share the GPU):
it proves the sharded path's outputs, it is not a measurement.  Without a GPU the command fails in engine creation
(there is no CPU path).  Reference: chiron_eval.py:277-285 (file walk), README.md:156 / utils/merge.sh (gather).
"""
import argparse
import filecmp
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_reads(folder, n_reads, n_samples, seed=1234, first=0, kind="signal"):
    """reads first .. first + n_reads - 1 as raw/<read>.signal files in the format extract_sig_ref.py:122-123 writes (one
    integer per line), or as fast5 files (chunked + deflate int16, tests/h5_writer.py)."""
    import numpy as np
    import chiron_amd as ca
    from chiron_amd import fast5
    os.makedirs(folder, exist_ok=True)
    if kind == "fast5":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import h5_writer
    for r in range(first, first + n_reads):
        sig = ca.synthetic_signal(1, n_samples + (r % 7) * 131, seed=seed + r)[0]     # ragged lengths: ragged last windows
        if kind == "fast5":
            h5_writer.write_multi_read_fast5(os.path.join(folder, "read%05d.fast5" % r), [("", "id-%d" % r, sig.astype(np.int16), None)], chunk=20000)
        else:
            fast5.write_signal_text(os.path.join(folder, "read%05d.signal" % r), sig, "\n")


def write_model_dir(folder):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "model.json"), "w") as f:
        json.dump({"cnn": {"model": "dna_model1"},
                   "rnn": {"layer_num": 3, "hidden_num": 100, "cell_type": "LSTM", "layer_type": "normal"}}, f)


def call(inp, out, model, ranks, share_gpu, extension="fastq", batch=1100, port=29611, timeout=3600, launcher="torchrun", dtype="fp32"):
    """`chiron call` (python -m chiron_amd.entry) on `ranks` processes; returns wall seconds.  launcher "torchrun": the ranks are
    started by torch.distributed.run (barriers over RCCL / gloo); "local": by `chiron call --gpus N` itself (file barrier, no torch)."""
    cmd = ["-m", "chiron_amd.entry", "call", "-i", inp, "-o", out, "-m", model, "--synthetic-weights", "-b", str(batch),
           "-l", "400", "-j", "390", "--beam", "0", "-e", extension, "-t", "4", "--dtype", dtype]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if ranks > 1 and launcher == "local":
        cmd += ["--gpus", str(ranks)]
        if share_gpu:
            env["CHIRON_SHARE_GPU"] = "1"
    elif ranks > 1:
        cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + cmd
        env["MASTER_ADDR"] = "127.0.0.1"
        if share_gpu:
            env["CHIRON_SHARE_GPU"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("chiron call on %d rank(s) failed:\n%s" % (ranks, r.stdout[-4000:]))
    return time.time() - t0


def compare_trees(a, b, extension):
    """result/ and segments/ of two output trees must hold the same files with the same bytes -> number compared."""
    n = 0
    for sub in ("result", "segments"):
        fa, fb = sorted(os.listdir(os.path.join(a, sub))), sorted(os.listdir(os.path.join(b, sub)))
        if fa != fb:
            raise AssertionError("%s/: file lists differ (%d vs %d files)" % (sub, len(fa), len(fb)))
        for name in fa:
            if not filecmp.cmp(os.path.join(a, sub, name), os.path.join(b, sub, name), shallow=False):
                raise AssertionError("%s/%s differs between the 1-rank and the sharded run" % (sub, name))
            n += 1
    return n


def run_chunk(workdir, first, n_reads, n_samples, ranks, share_gpu, extension, batch, kind, keep, launcher="torchrun", dtype="fp32"):
    """one chunk: inputs written, 1-rank and R-rank `chiron call`, trees compared, everything deleted again -> record"""
    import hashlib
    import shutil
    from chiron_amd import shard
    cdir = os.path.join(workdir, "chunk_%06d" % first)
    shutil.rmtree(cdir, ignore_errors=True)            # a chunk interrupted half way starts over
    sig, model = os.path.join(cdir, "input"), os.path.join(workdir, "model")
    write_reads(sig, n_reads, n_samples, first=first, kind=kind)
    write_model_dir(model)
    out1, outn = os.path.join(cdir, "out_1rank"), os.path.join(cdir, "out_%dranks" % ranks)
    t1 = call(sig, out1, model, 1, False, extension, batch, dtype=dtype)
    merged1, n1 = shard.gather_results(out1, extension)          # a single process does not gather by itself
    shutil.rmtree(os.path.join(out1, "raw"), ignore_errors=True)   # the largest folder; never compared
    tn = call(sig, outn, model, ranks, share_gpu, extension, batch, launcher=launcher, dtype=dtype)
    mergedn = os.path.join(outn, "merged." + extension)
    if n1 != n_reads:
        raise AssertionError("%d reads in, %d results out" % (n_reads, n1))
    compared = compare_trees(out1, outn, extension)
    if not filecmp.cmp(merged1, mergedn, shallow=False):
        raise AssertionError("merged.%s differs between the 1-rank and the %d-rank run" % (extension, ranks))
    text = open(merged1, "rb").read()
    bases = sum(len(l) for i, l in enumerate(text.decode().split("\n")) if i % (4 if extension == "fastq" else 2) == 1)
    rec = {"first_read": first, "reads": n_reads, "files_compared": compared, "merged_bytes": len(text), "merged_sha256": hashlib.sha256(text).hexdigest(),
           "consensus_bases": bases, "wall_s_1rank": round(t1, 2), "wall_s_ranks": round(tn, 2)}
    if not keep:
        shutil.rmtree(cdir, ignore_errors=True)
    return rec


def run(workdir, n_reads, n_samples, ranks, share_gpu, extension="fastq", batch=1100, chunk=0, kind="signal", keep=False, launcher="torchrun",
        dtype="fp32"):
    os.makedirs(workdir, exist_ok=True)
    chunk = chunk if chunk > 0 else n_reads
    state_path = os.path.join(workdir, "state.json")
    key = {"reads": n_reads, "samples_per_read": n_samples, "ranks": ranks, "extension": extension, "batch": batch, "chunk": chunk, "input": kind,
           "launcher": launcher, "dtype": dtype}
    state = {"key": key, "chunks": {}}
    if os.path.exists(state_path):
        old = json.load(open(state_path))
        if old.get("key") == key:
            state = old                       # resume: finished chunks are not repeated
    for first in range(0, n_reads, chunk):
        if str(first) in state["chunks"]:
            continue
        state["chunks"][str(first)] = run_chunk(workdir, first, min(chunk, n_reads - first), n_samples, ranks, share_gpu, extension, batch, kind, keep, launcher, dtype)
        tmp = state_path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(state, f, indent=1, sort_keys=True)
        os.replace(tmp, state_path)
    recs = [state["chunks"][k] for k in sorted(state["chunks"], key=int)]
    return {"reads": n_reads, "samples_per_read": n_samples, "ranks": ranks, "share_gpu": bool(share_gpu), "input": kind, "launcher": launcher,
            "chunks": len(recs),
            "files_compared": sum(r["files_compared"] for r in recs), "merged_bytes": sum(r["merged_bytes"] for r in recs),
            "consensus_bases": sum(r["consensus_bases"] for r in recs), "identical": True,
            "wall_s_1rank": round(sum(r["wall_s_1rank"] for r in recs), 2), "wall_s_%dranks" % ranks: round(sum(r["wall_s_ranks"] for r in recs), 2)}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--reads", type=int, default=24)
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--extension", default="fastq")
    ap.add_argument("--batch", type=int, default=1100)
    ap.add_argument("--workdir", default=None, help="kept between runs: state.json there makes the run resumable")
    ap.add_argument("--chunk", type=int, default=0, help="reads per chunk (0: all at once); the disk holds one chunk at a time")
    ap.add_argument("--input", default="signal", choices=["signal", "fast5"], help="input files: .signal text or fast5 (direct path)")
    ap.add_argument("--keep", action="store_true", help="keep the chunk folders (inputs and both output trees)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "fp16-w2", "fp32-split"])
    ap.add_argument("--launcher", default="torchrun", choices=["torchrun", "local"],
                    help="who starts the ranks: torch.distributed.run, or `chiron call --gpus N` itself (file barrier, no torch)")
    a = ap.parse_args()
    import tempfile
    wd = a.workdir or tempfile.mkdtemp(prefix="chiron_shard_")
    print(json.dumps(run(wd, a.reads, a.samples, a.ranks, a.share_gpu, a.extension, a.batch, a.chunk, a.input, a.keep, a.launcher, a.dtype)))


if __name__ == "__main__":
    main()
