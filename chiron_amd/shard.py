"""Per-read sharding across GPUs: reads are independent units, so each rank basecalls
its own reads and writes its own result files; the only cross-rank step is a host-side FASTA/FASTQ
gather (what the reference documents as `utils/merge.sh`, README.md:156).  No RCCL collective touches
the data path -- torch.distributed is used for the barrier only (backend nccl on GPUs, gloo on CPU)."""
import os


def init_distributed():
    """One process per GPU under torch.distributed.run: -> (dist or None, rank, world, device or None).  RCCL (backend
    "nccl") when every rank has its own GPU; with CHIRON_SHARE_GPU=1 -- the self-test of the N > 1 path on a box with
    one GPU -- all ranks use device 0 and gloo carries the barriers (two RCCL ranks cannot share a device).  Only
    barriers ever go through it."""
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
