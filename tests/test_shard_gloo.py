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
