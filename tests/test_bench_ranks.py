"""CPU: the multi-rank bookkeeping of bench.py under torch.distributed.run with world sizes 2, 4 and 8 (the launch line the
driver uses for the 1/2/4/8-GPU scaling runs), a stub in place of the engine (`--stub-engine MS`: collect() of rank r
sleeps MS + r milliseconds) and gloo in place of RCCL.  No 8-GPU node was available to any round so far, so the N > 1 code
path -- per-rank seeds, one rank per device ordinal, barrier + synchronize bracket, MAX over ranks of the clock, whole-job
aggregate, weak-scaling arithmetic -- must not be what fails when one appears."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_line_under_torchrun_with_a_stub_engine(world):
    ms, steps, rounds = 12.0, 6, 2
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps),
           "--rounds", str(rounds), "--host-rounds", "1", "--warmup", "2", "--slots", "3", "--stub-engine", str(ms)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == steps and d["rounds"] == rounds and d["timed_steps"] == steps * rounds
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["data"] == "stub" and d["unit"] == "kbases/s"
    # every rank ran timed_steps batches of 1100 windows; the clock is the MAX over ranks, i.e. the slowest stub (ms + world - 1
    # per collect) sets it: value = world * windows * 43.875 bases / that time
    t = d["timed_region_s"]
    slowest = steps * rounds * (ms + world - 1) * 1e-3
    assert slowest * 0.98 <= t <= slowest * 1.6 + 0.5, (t, slowest)
    assert abs(d["ms_per_step"] - 1e3 * t / (steps * rounds)) < 0.1     # both are rounded in the line
    want = world * steps * rounds * 1100 * (390 / (4000.0 / 450.0)) / 1000.0 / t
    assert abs(d["value"] - want) / want < 5e-3
    assert abs(d["extra"]["windows_per_s"] - world * steps * rounds * 1100 / t) / (world * steps * rounds * 1100 / t) < 5e-3
    # the per-rank counters were summed over ranks (one decoded base per window in the stub)
    assert d["extra"]["decoded_bases_total"] == world * steps * rounds * 1100
    h = d["extra"]["host_inclusive"]
    assert h["timed_steps"] == steps and h["kbases_per_s"] > 0
