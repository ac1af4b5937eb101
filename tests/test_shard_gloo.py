"""CPU, world_size 2 over gloo: the N>1 path.  Reads shard per rank with no data-path collective; the
only cross-rank steps are the barrier and the host-side FASTA gather."""
import os
import subprocess
import sys

import pytest

from chiron_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_disjoint_cover():
    files = ["r%02d.signal" % i for i in range(11)]
    for world in (1, 2, 3, 8):
        parts = [shard.partition_reads(files, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(files)
        assert all(set(a).isdisjoint(b) for i, a in enumerate(parts) for b in parts[i + 1:])
    sizes = {f: (i + 1) * 100 for i, f in enumerate(files)}
    parts = [shard.partition_reads(files, 2, r, sizes) for r in range(2)]
    loads = [sum(sizes[f] for f in p) for p in parts]
    assert sorted(sum(parts, [])) == sorted(files) and abs(loads[0] - loads[1]) <= 1100


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from chiron_amd import shard
dist.init_process_group("gloo")
class F: pass
F.input, F.output, F.extension, F.recursive = %(inp)r, %(out)r, "fasta", False
def fake_basecall(FLAGS, files):
    os.makedirs(os.path.join(FLAGS.output, "result"), exist_ok=True)
    for f in files:
        stem = os.path.splitext(f)[0]
        with open(os.path.join(FLAGS.output, "result", stem + ".fasta"), "w") as o:
            o.write(">%%s\nACGT%%d" %% (stem, dist.get_rank()))
    return files
mine, merged = shard.run_sharded(F, fake_basecall, dist)
print("RANK", dist.get_rank(), sorted(mine), merged)
dist.destroy_process_group()
'''


def test_two_ranks_gloo_shard_and_gather(tmp_path):
    inp = tmp_path / "raw"
    inp.mkdir()
    for i in range(5):
        (inp / ("read%d.signal" % i)).write_text("1\n" * (10 * (i + 1)))
    (inp / "notes.txt").write_text("x")
    out = tmp_path / "out"
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "inp": str(inp), "out": str(out)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    names = sorted(os.listdir(str(out / "result")))
    assert names == ["read%d.fasta" % i for i in range(5)]
    merged = open(str(out / "merged.fasta")).read().split("\n")
    assert [l for l in merged if l.startswith(">")] == [">read%d" % i for i in range(5)]
    ranks = {l[-1] for l in merged if l.startswith("ACGT")}
    assert ranks == {"0", "1"}                      # both ranks contributed reads


def _barrier_worker(args):
    rank, world, folder, fail = args
    import time
    from chiron_amd import shard as sh
    d = sh.LocalRanks(rank, world, folder, timeout_s=60)
    order = []
    for phase in range(3):
        time.sleep(0.01 * ((rank + phase) % world))          # ranks arrive in a different order every phase
        if fail and rank == 1 and phase == 1:
            open(os.path.join(folder, "failed.1"), "w").close()      # what spawn_local_ranks writes for a rank that died
            return ("died", rank)
        try:
            d.barrier()
        except RuntimeError as e:
            return ("released", rank, str(e))
        order.append(sorted(n for n in os.listdir(folder) if n.startswith("barrier.%d." % (phase + 1))))
    return ("ok", rank, order)


def test_file_barrier_of_the_self_spawned_ranks(tmp_path):
    """`chiron call --gpus N` starts its ranks itself; their only shared step is a barrier carried by marker files
    (shard.LocalRanks): nobody passes phase p before every rank has reached it, three phases, ranks arriving in changing order;
    a rank that died (failed.<rank>) releases the others with an error instead of a hang; rank_cpus cuts the host's cores into
    contiguous, disjoint slices that cover them."""
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    world = 4
    with ctx.Pool(world) as pool:
        res = pool.map(_barrier_worker, [(r, world, str(tmp_path / "ok"), False) for r in range(world)])
    assert [r[0] for r in res] == ["ok"] * world
    for _, rank, order in res:
        for phase, seen in enumerate(order):
            assert seen == ["barrier.%d.%d" % (phase + 1, r) for r in range(world)]      # all markers present when anyone leaves
    with ctx.Pool(world) as pool:
        res = pool.map(_barrier_worker, [(r, world, str(tmp_path / "fail"), True) for r in range(world)])
    assert sorted(r[0] for r in res) == ["died", "released", "released", "released"]
    assert all("rank(s) 1 failed" in r[2] for r in res if r[0] == "released")
    for n, world in ((128, 8), (8, 8), (10, 4), (3, 8)):
        parts = [shard.rank_cpus(r, world, range(n)) for r in range(world)]
        if n >= world:
            assert sum(parts, []) == list(range(n)) and all(parts) and max(map(len, parts)) - min(map(len, parts)) <= 1
        else:
            assert all(p == list(range(n)) for p in parts)        # fewer cores than ranks: no pinning


def test_call_with_gpus_fails_loudly_and_does_not_hang_without_a_gpu(tmp_path):
    """`chiron call --gpus 2` on a box without a GPU: both rank processes fail in engine creation (there is no CPU path), the
    parent reports their exit codes -- within seconds, nobody waits at a barrier for a dead rank."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    inp = tmp_path / "raw"
    inp.mkdir()
    (inp / "r.signal").write_text("\n".join(["500"] * 1000))
    model = os.path.join(ROOT, "chiron_amd", "model", "DNA_default")
    r = subprocess.run([sys.executable, "-m", "chiron_amd.entry", "call", "-i", str(inp), "-o", str(tmp_path / "out"), "-m", model,
                        "--synthetic-weights", "-p", "dna-pre", "-b", "16", "--beam", "0", "--gpus", "2"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=120)
    assert r.returncode != 0 and "rank exit codes" in r.stdout and "no CPU fallback" in r.stdout, r.stdout[-1500:]
    assert sorted(os.listdir(str(tmp_path / "out" / "log" / "ranks"))) == ["failed.0", "failed.1"]
