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
    # round-5 review item 5: the bookkeeping runs over gloo by default (no RCCL bring-up between a node and its first record) and the
    # record names every rank's device
    assert d["config"]["parallelism_bookkeeping"].startswith("gloo (cpu tensors): ")
    assert d["config"]["devices"] == ["stub:%d" % r for r in range(world)]


def _bench(argv, env=None, timeout=600):
    e = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_bare_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 4` WITHOUT a launcher (the way the driver's 1-GPU line is started) must not run one rank and print
    "n_gpus": 1 with rc 0 (round-4 review, Missing #3): it starts four ranks itself and rank 0's line says n_gpus 4."""
    out = _bench(["--gpus", "4", "--stub-engine", "12", "--steps", "4", "--rounds", "2", "--host-rounds", "0", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["timed_steps"] == 8 and d["extra"]["decoded_bases_total"] == 4 * 8 * 1100


def test_backend_nccl_falls_back_to_gloo_when_rccl_does_not_come_up():
    """`--backend nccl` on a box without a GPU: creating the RCCL group raises on every rank; the ranks agree over the gloo default
    group, stay on it, and the line says which backend carried the bookkeeping."""
    out = _bench(["--gpus", "2", "--stub-engine", "8", "--backend", "nccl", "--steps", "3", "--rounds", "1", "--host-rounds", "0", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["extra"]["decoded_bases_total"] == 2 * 3 * 1100
    assert d["config"]["parallelism_bookkeeping"].startswith("gloo (cpu tensors) after RCCL bring-up failed on 2 of 2 ranks")
    assert "bookkeeping stays on gloo" in out.stderr


def test_bench_refuses_a_launcher_whose_world_size_is_not_gpus():
    out = _bench(["--gpus", "4", "--stub-engine", "12"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert out.returncode == 2 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = _bench(["--gpus", "1", "--stub-engine", "12"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert out.returncode == 2


def test_ranks_must_sit_on_distinct_devices():
    sys.path.insert(0, ROOT)
    import bench
    hw = lambda i: ("uuid%d/0/%d/0" % (i, i), i)
    assert not bench.ranks_share_a_device([hw(i) for i in range(8)], 8)
    assert bench.ranks_share_a_device([hw(0), hw(1), hw(1), hw(3)], 8)                     # two ranks on GPU 1
    assert bench.ranks_share_a_device([hw(0)] * 8, 8)                                      # eight ranks on GPU 0
    same = [("0000/0/0/0", i) for i in range(8)]                                           # a runtime that fills nothing in
    assert not bench.ranks_share_a_device(same, 8) and bench.ranks_share_a_device(same, 4)
    assert not bench.ranks_share_a_device(["stub:0", "stub:1"], 2) and bench.ranks_share_a_device(["stub:0", "stub:0"], 2)
    # two ranks given the same ordinal under a launcher: refused unless it is the declared self-test
    e = {"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine", "5", "--steps", "2", "--rounds", "1",
                            "--host-rounds", "0", "--warmup", "1"], cwd=ROOT, env=dict(os.environ, RANK=str(r), LOCAL_RANK="0", **e),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in ps]
    assert [p.returncode for p in ps] == [4, 4] and "share a device" in outs[0][1]


def test_live_pmc_counter_files_to_hbm_bytes(tmp_path):
    """bench.py measures roofline.traffic in the run itself: two child passes under rocprofv3 --pmc write *counter_collection.csv files;
    this is the parsing half (the GPU half runs with the bench): kernel names lose `void`, the namespace and the argument list, values
    are averaged per launch, hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, a kernel seen in only one pass is left out."""
    sys.path.insert(0, ROOT)
    import bench
    head = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    row = '1,1,1,1,1,1,1,1,"%s",256,0,0,64,16,%s,%s,0,1\n'
    rd, wr = tmp_path / "a_counter_collection.csv", tmp_path / "b_counter_collection.csv"
    rd.write_text(head + row % ("chiron::lstm32w2_kernel(chiron::LstmParams)", "FETCH_SIZE", "1000") + row % ("chiron::lstm32w2_kernel(chiron::LstmParams)", "FETCH_SIZE", "3000")
                  + row % ("void chiron::gemm_f32_dma_kernel<true, false, 8, false, 0, true>(chiron::GemmParams)", "FETCH_SIZE", "10") + row % ("only_read(int)", "FETCH_SIZE", "5"))
    wr.write_text(head + row % ("chiron::lstm32w2_kernel(chiron::LstmParams)", "WRITE_SIZE", "500")
                  + row % ("void chiron::gemm_f32_dma_kernel<true, false, 8, false, 0, true>(chiron::GemmParams)", "WRITE_SIZE", "20"))
    acc = bench.collect_counter([str(rd)], "FETCH_SIZE", {})
    acc = bench.collect_counter([str(wr)], "WRITE_SIZE", acc)
    k = bench.hbm_bytes_per_launch(acc)
    assert set(k) == {"lstm32w2_kernel", "gemm_f32_dma_kernel<true, false, 8, false, 0, true>"}
    assert k["lstm32w2_kernel"]["hbm_bytes"] == (2 * 2000 + 500) * 1024 and k["lstm32w2_kernel"]["launches_fetch"] == 2
    assert k["gemm_f32_dma_kernel<true, false, 8, false, 0, true>"]["hbm_bytes"] == (2 * 10 + 20) * 1024
    assert any(bench.BUCKET_SYMBOL["lstm_proj0_dma"] in name for name in k)       # the bucket -> symbol table finds its kernel by substring
