"""Per-read sharding across GPUs: reads are independent units, so each rank basecalls
its own reads and writes its own result files; the only cross-rank step is a host-side FASTA/FASTQ
gather (what the reference documents as `utils/merge.sh`, README.md:156).  No RCCL collective touches
the data path -- the ranks share a barrier and nothing else: torch.distributed's when the processes were started by
torch.distributed.run (backend nccl on GPUs, gloo on CPU), a folder of marker files when `chiron call --gpus N` started
them itself (LocalRanks; no torch import, no port)."""
import os
import time


class LocalRanks(object):
    """The ranks of `chiron call --gpus N` (spawn_local_ranks below): the data path needs no collective, so the only thing the
    ranks share is a BARRIER, and a folder of marker files carries it -- no torch.distributed, no RCCL, no port.  Same surface
    as the three torch.distributed calls run_sharded / entry use (get_rank, get_world_size, barrier, destroy_process_group)."""

    def __init__(self, rank, world, folder, timeout_s=24 * 3600.0):
        self.rank, self.world, self.folder, self.timeout_s = int(rank), int(world), folder, timeout_s
        self.phase = 0
        os.makedirs(folder, exist_ok=True)

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def barrier(self):
        """every rank writes barrier.<phase>.<rank> and waits until all `world` markers of the phase exist; a rank that died
        leaves `failed.<rank>` (spawn_local_ranks' parent writes it) and the others stop waiting."""
        self.phase += 1
        mine = os.path.join(self.folder, "barrier.%d.%d" % (self.phase, self.rank))
        with open(mine, "w") as f:
            f.write("%d\n" % os.getpid())
        want = ["barrier.%d.%d" % (self.phase, r) for r in range(self.world)]
        t0 = time.time()
        delay = 0.002
        while True:
            have = set(os.listdir(self.folder))
            if all(w in have for w in want):
                return
            dead = sorted(n for n in have if n.startswith("failed."))
            if dead:
                raise RuntimeError("rank(s) %s failed: the barrier cannot complete" % ", ".join(n.split(".")[1] for n in dead))
            if time.time() - t0 > self.timeout_s:
                raise RuntimeError("barrier %d timed out after %.0f s" % (self.phase, self.timeout_s))
            time.sleep(delay)
            delay = min(0.05, delay * 1.5)

    def destroy_process_group(self):
        pass


def rank_cpus(rank, world, cpus=None):
    """CPU affinity of one local rank: the r-th of `world` contiguous slices of the cores this process may use (a rank's reader
    and finisher threads then stay on cores next to each other -- one L3 / NUMA neighbourhood on the usual enumeration --
    instead of migrating over the whole host between eight ranks' threads)."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    n = len(cpus)
    if world <= 1 or n < world:
        return cpus
    lo, hi = rank * n // world, (rank + 1) * n // world
    return cpus[lo:hi]


def spawn_local_ranks(argv, n_gpus, output, python=None, share_gpu=False):
    """`chiron call --gpus N`: N child processes of `python -m chiron_amd.entry <argv>`, rank r on GPU r (on GPU 0 with
    share_gpu: the self-test on a one-GPU box) and on its own slice of the host's cores; CHIRON_LOCAL_RANK / CHIRON_LOCAL_WORLD /
    CHIRON_BARRIER_DIR tell the child who it is (init_distributed below).  -> list of exit codes (the caller raises)."""
    import shutil
    import subprocess
    import sys
    folder = os.path.join(output, "log", "ranks")
    shutil.rmtree(folder, ignore_errors=True)
    os.makedirs(folder, exist_ok=True)
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, CHIRON_LOCAL_RANK=str(r), CHIRON_LOCAL_WORLD=str(n_gpus), CHIRON_BARRIER_DIR=folder,
                   LOCAL_WORLD_SIZE=str(n_gpus))
        env.pop("WORLD_SIZE", None)          # the children are not torch.distributed ranks
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        if share_gpu:
            env["CHIRON_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([python or sys.executable, "-m", "chiron_amd.entry"] + list(argv), env=env))
    codes = [None] * n_gpus
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None and p.poll() is not None:
                codes[r] = p.returncode
                if p.returncode != 0:        # let the others leave their barrier instead of waiting for a dead rank
                    open(os.path.join(folder, "failed.%d" % r), "w").close()
        time.sleep(0.02)
    return codes


def init_distributed():
    """One process per GPU under torch.distributed.run: -> (dist or None, rank, world, device or None).  RCCL (backend
    "nccl") when every rank has its own GPU; with CHIRON_SHARE_GPU=1 -- the self-test of the N > 1 path on a box with
    one GPU -- all ranks use device 0 and gloo carries the barriers (two RCCL ranks cannot share a device).  Only
    barriers ever go through it."""
    if os.environ.get("CHIRON_LOCAL_WORLD"):          # a child of `chiron call --gpus N`: file barrier, no torch
        rank, world = int(os.environ["CHIRON_LOCAL_RANK"]), int(os.environ["CHIRON_LOCAL_WORLD"])
        try:
            os.sched_setaffinity(0, rank_cpus(rank, world))
        except (AttributeError, OSError):
            pass
        device = 0 if os.environ.get("CHIRON_SHARE_GPU") == "1" else rank
        return LocalRanks(rank, world, os.environ["CHIRON_BARRIER_DIR"]), rank, world, device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("CHIRON_SHARE_GPU") == "1"
    if torch.cuda.is_available() and not share:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        device = local
    else:
        dist.init_process_group("gloo")
        device = 0
    return dist, dist.get_rank(), dist.get_world_size(), device


def partition_reads(files, world_size, rank, sizes=None):
    """Deterministic partition of the (sorted) read list.  Without sizes: read k -> rank k mod G.
    With sizes (bytes or samples): greedy longest-first balancing, ties by name."""
    files = sorted(files)
    if world_size <= 1:
        return files
    if sizes is None:
        return [f for k, f in enumerate(files) if k % world_size == rank]
    load = [0] * world_size
    owner = {}
    for f in sorted(files, key=lambda x: (-sizes[x], x)):
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[f] = r
        load[r] += sizes[f]
    return [f for f in files if owner[f] == rank]


def gather_results(output_dir, extension="fastq", merged_name="merged"):
    """Concatenate result/<read>.<ext> (sorted by name) into <output>/<merged_name>.<ext>."""
    res = os.path.join(output_dir, "result")
    names = sorted(n for n in os.listdir(res) if n.endswith("." + extension))
    out_path = os.path.join(output_dir, merged_name + "." + extension)
    with open(out_path, "w") as out:
        for n in names:
            txt = open(os.path.join(res, n)).read()
            out.write(txt if txt.endswith("\n") else txt + "\n")
    return out_path, len(names)


def run_sharded(FLAGS, basecall_fn, dist=None, partition=True):
    """One process per GPU.  basecall_fn(FLAGS, file_list) handles this rank's reads.
    `dist` is torch.distributed (already initialised) or None for a single process.  partition=False: the caller has
    already assigned this rank its inputs (the direct fast5 path partitions the fast5 files themselves)."""
    from . import eval as chiron_eval
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = None
    if partition:
        files, file_dir = chiron_eval.list_inputs(FLAGS.input, getattr(FLAGS, "recursive", False))
        files = [f for f in files if f.endswith(".signal") or f.endswith(".fast5")]
        sizes = {f: os.path.getsize(os.path.join(file_dir, f)) for f in files}
        mine = partition_reads(files, world, rank, sizes)
    out = basecall_fn(FLAGS, mine)
    if dist is not None:
        dist.barrier()
    merged = None
    if rank == 0:
        merged = gather_results(FLAGS.output, FLAGS.extension)
    if dist is not None:
        dist.barrier()
    return out, merged
