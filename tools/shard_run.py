#!/usr/bin/env python3
"""BASELINE configs[3] end to end: N synthetic reads x S samples -> `chiron call` on 1 rank and on R ranks
(one process per GPU under torch.distributed.run, reads sharded per rank, host-side gather) -> the merged FASTA/FASTQ
and every result file must be byte-identical (meta/ holds timings and is not compared).

    python tools/shard_run.py --reads 10000 --samples 100000 --ranks 8            # the configuration as written
    python tools/shard_run.py --reads 24 --ranks 2 --share-gpu                    # self-test on a 1-GPU box

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


def write_reads(folder, n_reads, n_samples, seed=1234):
    """raw/<read>.signal files in the format extract_sig_ref.py:122-123 writes (one integer per line)."""
    import chiron_amd as ca
    os.makedirs(folder, exist_ok=True)
    for r in range(n_reads):
        sig = ca.synthetic_signal(1, n_samples + (r % 7) * 131, seed=seed + r)[0]     # ragged lengths: ragged last windows
        with open(os.path.join(folder, "read%05d.signal" % r), "w") as f:
            f.write("\n".join(str(int(v)) for v in sig))


def write_model_dir(folder):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "model.json"), "w") as f:
        json.dump({"cnn": {"model": "dna_model1"},
                   "rnn": {"layer_num": 3, "hidden_num": 100, "cell_type": "LSTM", "layer_type": "normal"}}, f)


def call(inp, out, model, ranks, share_gpu, extension="fastq", batch=1100, port=29611, timeout=3600):
    """`chiron call` (python -m chiron_amd.entry) on `ranks` processes; returns wall seconds."""
    cmd = ["-m", "chiron_amd.entry", "call", "-i", inp, "-o", out, "-m", model, "--synthetic-weights", "-b", str(batch),
           "-l", "400", "-j", "390", "--beam", "0", "-e", extension, "-t", "4"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if ranks > 1:
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


def run(workdir, n_reads, n_samples, ranks, share_gpu, extension="fastq", batch=1100):
    from chiron_amd import shard
    sig, model = os.path.join(workdir, "signals"), os.path.join(workdir, "model")
    write_reads(sig, n_reads, n_samples)
    write_model_dir(model)
    out1, outn = os.path.join(workdir, "out_1rank"), os.path.join(workdir, "out_%dranks" % ranks)
    t1 = call(sig, out1, model, 1, False, extension, batch)
    merged1, n1 = shard.gather_results(out1, extension)          # a single process does not gather by itself
    tn = call(sig, outn, model, ranks, share_gpu, extension, batch)
    mergedn = os.path.join(outn, "merged." + extension)
    if n1 != n_reads:
        raise AssertionError("%d reads in, %d results out" % (n_reads, n1))
    compared = compare_trees(out1, outn, extension)
    if not filecmp.cmp(merged1, mergedn, shallow=False):
        raise AssertionError("merged.%s differs between the 1-rank and the %d-rank run" % (extension, ranks))
    bases = sum(len(l) for i, l in enumerate(open(merged1).read().split("\n")) if i % (4 if extension == "fastq" else 2) == 1)
    return {"reads": n_reads, "samples_per_read": n_samples, "ranks": ranks, "share_gpu": bool(share_gpu), "files_compared": compared,
            "merged_bytes": os.path.getsize(merged1), "consensus_bases": bases, "identical": True,
            "wall_s_1rank": round(t1, 2), "wall_s_%dranks" % ranks: round(tn, 2)}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--reads", type=int, default=24)
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--extension", default="fastq")
    ap.add_argument("--batch", type=int, default=1100)
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()
    import tempfile
    wd = a.workdir or tempfile.mkdtemp(prefix="chiron_shard_")
    print(json.dumps(run(wd, a.reads, a.samples, a.ranks, a.share_gpu, a.extension, a.batch)))


if __name__ == "__main__":
    main()
