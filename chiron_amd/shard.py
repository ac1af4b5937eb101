"""Per-read sharding across GPUs: reads are independent units, so each rank basecalls
its own reads and writes its own result files; the only cross-rank step is a host-side FASTA/FASTQ
gather (what the reference documents as `utils/merge.sh`, README.md:156).  No RCCL collective touches
the data path -- the ranks share a barrier and nothing else: torch.distributed's when the processes were started by
torch.distributed.run (backend nccl on GPUs, gloo on CPU), a folder of marker files when `chiron call --gpus N` started
them itself (LocalRanks; no torch import, no port)."""
import os
import time


def _alive(pid):
    try:
        os.kill(int(pid), 0)
    except (OSError, ValueError):
        return False
    try:                                       # a zombie still answers kill(pid, 0): it has exited
        with open("/proc/%d/stat" % int(pid)) as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except (OSError, IndexError):
        return True


class LocalRanks(object):
    """The ranks of `chiron call --gpus N` (spawn_local_ranks below): the data path needs no collective, so the only thing the
    ranks share is a BARRIER, and a folder of marker files carries it -- no torch.distributed, no RCCL, no port.  Same surface
    as the three torch.distributed calls run_sharded / entry use (get_rank, get_world_size, barrier, destroy_process_group).
    A waiting rank gives up -- and releases its GPU -- when (a) the parent left failed.<r> for a rank that exited non-zero, (b) the
    parent itself is gone (SIGKILL / OOM: nobody would ever write failed.*), (c) a peer whose pid is on record (pid.<r>, written by
    every rank at start-up) no longer exists, or (d) CHIRON_BARRIER_TIMEOUT_S (default one hour: the longest a rank may lag behind the
    others, not the length of the job) has passed."""

    def __init__(self, rank, world, folder, timeout_s=None, parent_pid=None):
        self.rank, self.world, self.folder = int(rank), int(world), folder
        self.timeout_s = float(os.environ.get("CHIRON_BARRIER_TIMEOUT_S", 3600.0)) if timeout_s is None else float(timeout_s)
        self.parent_pid = parent_pid
        self.phase = 0
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "pid.%d" % self.rank), "w") as f:
            f.write("%d\n" % os.getpid())

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def _peers_gone(self):
        gone = []
        for r in range(self.world):
            if r == self.rank:
                continue
            try:
                with open(os.path.join(self.folder, "pid.%d" % r)) as f:
                    pid = f.read().strip()
            except OSError:
                continue                       # not started yet
            if pid and not _alive(pid):
                gone.append(r)
        return gone

    def barrier(self):
        """every rank writes barrier.<phase>.<rank> and waits until all `world` markers of the phase exist"""
        self.phase += 1
        mine = os.path.join(self.folder, "barrier.%d.%d" % (self.phase, self.rank))
        with open(mine, "w") as f:
            f.write("%d\n" % os.getpid())
        want = ["barrier.%d.%d" % (self.phase, r) for r in range(self.world)]
        t0 = time.time()
        delay, checked = 0.002, t0
        while True:
            have = set(os.listdir(self.folder))
            if all(w in have for w in want):
                return
            dead = sorted(n for n in have if n.startswith("failed."))
            if dead:
                raise RuntimeError("rank(s) %s failed: the barrier cannot complete" % ", ".join(n.split(".")[1] for n in dead))
            now = time.time()
            if now - checked > 0.5:            # liveness twice a second: a kill(pid, 0) per peer
                checked = now
                if self.parent_pid and not _alive(self.parent_pid):
                    raise RuntimeError("the process that started the ranks (pid %s) is gone: leaving barrier %d" % (self.parent_pid, self.phase))
                gone = self._peers_gone()
                if gone:
                    # a peer may have written its marker, passed the barrier and exited (cleanly) between the listing above and the
                    # kill(pid, 0) probes: list again before calling it dead
                    have = set(os.listdir(self.folder))
                    gone = [r for r in gone if "barrier.%d.%d" % (self.phase, r) not in have]
                if gone:
                    raise RuntimeError("rank(s) %s exited before barrier %d" % (", ".join(map(str, gone)), self.phase))
            if now - t0 > self.timeout_s:
                raise RuntimeError("barrier %d timed out after %.0f s (CHIRON_BARRIER_TIMEOUT_S)" % (self.phase, self.timeout_s))
            time.sleep(delay)
            delay = min(0.05, delay * 1.5)

    def destroy_process_group(self):
        pass


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(world, share_gpu=False, sysfs="/sys"):
    """NUMA node of the GPU each local rank computes on, or None where it cannot be told: HIP device r -> its PCI address
    (chiron_device_pci_bus_id: the runtime's ordinal, not /sys/class/drm's card numbering, which also counts other adapters) ->
    <sysfs>/bus/pci/devices/<address>/numa_node (-1 on a single-node host)."""
    from . import _lib
    import ctypes as C
    try:
        lib = _lib.load()
    except (ImportError, OSError):
        return [None] * world
    nodes = []
    for r in range(world):
        buf = C.create_string_buffer(64)
        node = None
        if lib.chiron_device_pci_bus_id(0 if share_gpu else r, buf, 64) == 0:
            try:
                with open(os.path.join(sysfs, "bus", "pci", "devices", buf.value.decode(), "numa_node")) as f:
                    v = int(f.read().strip())
                node = v if v >= 0 else None
            except (OSError, ValueError):
                node = None
        nodes.append(node)
    return nodes


def node_cpus(sysfs="/sys"):
    """{NUMA node: [cpu, ...]} of this host"""
    base = os.path.join(sysfs, "devices", "system", "node")
    out = {}
    try:
        for name in os.listdir(base):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(base, name, "cpulist")) as f:
                    out[int(name[4:])] = _parse_cpulist(f.read())
    except OSError:
        pass
    return out


def core_order(cpus, sysfs="/sys"):
    """`cpus` re-ordered so that the hardware threads of one physical core are neighbours (cores in ascending order of their first
    thread).  Linux numbers the second SMT thread of core k as k + <cores of the host> (EPYC 9575F x 2: cpu 0 and cpu 128 are one
    core), so contiguous slices of the plain numbering gave rank 0 the first threads of cores 0..31 and rank 4 their siblings: two
    ranks' reader threads on the same cores' L1 / L2.  Sliced in THIS order a rank gets whole cores."""
    cpus = sorted(cpus)
    first = {}
    for c in cpus:
        key = c
        try:
            with open(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology", "thread_siblings_list")) as f:
                sib = _parse_cpulist(f.read())
            if sib:
                key = min(sib)
        except (OSError, ValueError):
            pass
        first[c] = key
    return sorted(cpus, key=lambda c: (first[c], c))


def rank_affinity(rank, world, share_gpu=False, sysfs="/sys"):
    """The cores `chiron call --gpus N` pins rank `rank` to: the NUMA node of its GPU, whole physical cores (rank_cpus + core_order)."""
    allowed = core_order(os.sched_getaffinity(0), sysfs)
    nodes = {n: [c for c in core_order(v, sysfs)] for n, v in node_cpus(sysfs).items()}
    return rank_cpus(rank, world, allowed, gpu_numa_nodes(world, share_gpu, sysfs), nodes, ordered=True)


def rank_cpus(rank, world, cpus=None, gpu_nodes=None, cpus_of_node=None, ordered=False):
    """CPU affinity of one local rank.  With the NUMA node of every rank's GPU known (gpu_nodes[r], gpu_numa_nodes) and the
    host's node -> cpus map: the ranks whose GPUs hang off one node share THAT node's allowed cores in contiguous slices -- a rank's
    pinned staging buffers, its reader / finisher threads and its GPU's PCIe root then sit on one node (round-4 review, Weak #9: the
    slices used to go by rank index alone).  Without that knowledge (single-node host, no sysfs, no GPU): the r-th of `world`
    contiguous slices of the allowed cores, as before."""
    if not ordered:          # ordered: the caller's lists are already in the order to slice (core_order)
        cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    n = len(cpus)
    if world <= 1 or n < world:
        return cpus
    if gpu_nodes and cpus_of_node and len(gpu_nodes) == world and all(g is not None for g in gpu_nodes):
        node = gpu_nodes[rank]
        allowed = set(cpus)
        mine = [c for c in cpus_of_node.get(node, []) if c in allowed]
        sharers = [r for r in range(world) if gpu_nodes[r] == node]
        if len(mine) >= len(sharers):
            i = sharers.index(rank)
            return mine[i * len(mine) // len(sharers):(i + 1) * len(mine) // len(sharers)]
    lo, hi = rank * n // world, (rank + 1) * n // world
    return cpus[lo:hi]


def spawn_local_ranks(argv, n_gpus, output, python=None, share_gpu=False, grace_s=10.0):
    """`chiron call --gpus N`: N child processes of `python -m chiron_amd.entry <argv>`, rank r on GPU r (on GPU 0 with
    share_gpu: the self-test on a one-GPU box) and on the cores next to its GPU; CHIRON_LOCAL_RANK / CHIRON_LOCAL_WORLD /
    CHIRON_BARRIER_DIR (+ CHIRON_PARENT_PID) tell the child who it is (init_distributed below).  When a rank exits non-zero the
    others are told (failed.<r>: they leave their next barrier) and, `grace_s` seconds later, terminated if they are still running
    -- a rank in the middle of a long basecall must not hold its GPU for a job that has already failed.
    -> list of exit codes (the caller raises)."""
    import shutil
    import subprocess
    import sys
    folder = os.path.join(output, "log", "ranks")
    shutil.rmtree(folder, ignore_errors=True)
    os.makedirs(folder, exist_ok=True)
    procs = []
    # placement is decided ONCE, here: the GPU -> NUMA node map costs a library load and N PCI lookups, and a rank that pinned itself
    # after importing the library would leave the threads created during that import on the whole host.  Each child is pinned between
    # fork and exec, so everything it ever starts inherits its cores; CHIRON_RANK_CPUS records the list (log/engine*.json).
    try:
        cpus = [rank_affinity(r, n_gpus, share_gpu) for r in range(n_gpus)]
    except (AttributeError, OSError):
        cpus = [None] * n_gpus

    def pin(c):
        def f():
            try:
                os.sched_setaffinity(0, c)
            except (AttributeError, OSError):
                pass
        return f if c else None

    for r in range(n_gpus):
        env = dict(os.environ, CHIRON_LOCAL_RANK=str(r), CHIRON_LOCAL_WORLD=str(n_gpus), CHIRON_BARRIER_DIR=folder,
                   CHIRON_PARENT_PID=str(os.getpid()), LOCAL_WORLD_SIZE=str(n_gpus))
        if cpus[r]:
            env["CHIRON_RANK_CPUS"] = ",".join(map(str, cpus[r]))
        env.pop("WORLD_SIZE", None)          # the children are not torch.distributed ranks
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        if share_gpu:
            env["CHIRON_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([python or sys.executable, "-m", "chiron_amd.entry"] + list(argv), env=env, preexec_fn=pin(cpus[r])))
    codes = [None] * n_gpus
    failed_at = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None and p.poll() is not None:
                codes[r] = p.returncode
                if p.returncode != 0:        # let the others leave their barrier instead of waiting for a dead rank
                    open(os.path.join(folder, "failed.%d" % r), "w").close()
                    failed_at = failed_at or time.time()
        if failed_at is not None and time.time() - failed_at > grace_s:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            failed_at = float("inf")         # once
        time.sleep(0.02)
    return codes


def init_distributed():
    """One process per GPU under torch.distributed.run: -> (dist or None, rank, world, device or None).  Only the two barriers
    of run_sharded ever go through the process group, so it is gloo (north_star: "no RCCL collectives" -- and a sharded call must
    not depend on RCCL bring-up); CHIRON_DIST_BACKEND=nccl selects RCCL.  With CHIRON_SHARE_GPU=1 -- the self-test of the N > 1
    path on a box with one GPU -- all ranks use device 0."""
    if os.environ.get("CHIRON_LOCAL_WORLD"):          # a child of `chiron call --gpus N`: file barrier, no torch
        rank, world = int(os.environ["CHIRON_LOCAL_RANK"]), int(os.environ["CHIRON_LOCAL_WORLD"])
        share = os.environ.get("CHIRON_SHARE_GPU") == "1"
        if not os.environ.get("CHIRON_RANK_CPUS"):      # started by something other than spawn_local_ranks (which pins before exec)
            try:
                os.sched_setaffinity(0, rank_affinity(rank, world, share))
            except (AttributeError, OSError):
                pass
        device = 0 if share else rank
        return LocalRanks(rank, world, os.environ["CHIRON_BARRIER_DIR"], parent_pid=os.environ.get("CHIRON_PARENT_PID")), rank, world, device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("CHIRON_SHARE_GPU") == "1"
    if torch.cuda.is_available() and not share:
        torch.cuda.set_device(local)
        if os.environ.get("CHIRON_DIST_BACKEND", "gloo") == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
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
