"""Command line (counterpart of chiron/entry.py): `python -m chiron_amd.entry call ...`.

Keeps the reference's `chiron call` flags and presets (entry.py:19-47, :69-92); only inference is
built (export/train are out of scope)."""
import argparse
import sys
from os import path

from . import __version__


def set_paras(args, p):
    """entry.py:53-60: explicit flags win over the preset."""
    args.start = p["start"] if args.start is None else args.start
    args.batch_size = p["batch_size"] if args.batch_size is None else args.batch_size
    args.segment_len = p["segment_len"] if args.segment_len is None else args.segment_len
    args.jump = p["jump"] if args.jump is None else args.jump
    args.threads = p["threads"] if args.threads is None else args.threads
    args.beam = p["beam"] if args.beam is None else args.beam
    return args


def resolve_preset(args):
    """entry.py:20-32."""
    if args.preset is None:
        default_p = {"start": 0, "batch_size": 400, "segment_len": 500, "jump": 490, "threads": 0, "beam": 30}
    elif args.preset == "dna-pre":
        default_p = {"start": 0, "batch_size": 400, "segment_len": 400, "jump": 390, "threads": 0, "beam": 30}
        if args.mode == "rna":
            raise ValueError("Try to use the DNA preset parameter setting in RNA mode.")
    elif args.preset == "rna-pre":
        default_p = {"start": 0, "batch_size": 300, "segment_len": 2000, "jump": 1900, "threads": 0, "beam": 30}
        if args.mode == "dna":
            raise ValueError("Attempt to use the RNA preset parameter setting in DNA mode, enable RNA basecalling by --mode rna")
    else:
        raise ValueError("Unknown presetting %s undifiend" % (args.preset))
    return set_paras(args, default_p)


def evaluation(args):
    """entry.py:19-47: extract fast5 -> <out>/raw, then basecall <out>/raw."""
    import os as _os
    n_gpus = int(getattr(args, "gpus", 0) or 0)
    child_vars = [v for v in ("CHIRON_LOCAL_RANK", "CHIRON_LOCAL_WORLD", "CHIRON_BARRIER_DIR") if _os.environ.get(v)]
    if child_vars and len(child_vars) != 3:
        # a rank of `chiron call --gpus N` gets all three from its parent; one or two of them is a stale export, and treating
        # the process as a rank would die later with a KeyError (or, worse, skip the spawn silently)
        raise RuntimeError("%s set without the other CHIRON_LOCAL_* variables: unset it (only `chiron call --gpus N` sets them, "
                           "all three, for its own ranks)" % ", ".join(child_vars))
    if n_gpus > 1 and not child_vars:
        # `chiron call --gpus N`: this process only starts the N ranks (one per GPU, each on its own slice of the host's cores)
        # and waits; the ranks shard the reads, meet at a file barrier and rank 0 gathers merged.<ext> (shard.py)
        from . import shard
        resolve_preset(args)                       # a bad preset / mode fails here, once, not N times
        _os.makedirs(args.output, exist_ok=True)
        child_argv = getattr(args, "child_argv", None)
        if child_argv is None:
            # evaluation(args) called with a Namespace instead of through main(): the ranks need a command line
            raise ValueError("--gpus %d needs the command line to give its ranks: call chiron_amd.entry.main([...]) or set "
                             "args.child_argv" % n_gpus)
        codes = shard.spawn_local_ranks(child_argv, n_gpus, args.output, share_gpu=_os.environ.get("CHIRON_SHARE_GPU") == "1")
        if any(codes):
            raise RuntimeError("chiron call --gpus %d: rank exit codes %s" % (n_gpus, codes))
        return codes
    from . import eval as chiron_eval
    from .extract import extract
    args = resolve_preset(args)
    FLAGS = args
    FLAGS.input_dir = FLAGS.input
    FLAGS.output_dir = FLAGS.output
    FLAGS.unit = False
    FLAGS.recursive = True
    FLAGS.polya = None
    FLAGS.idname = False
    FLAGS.delimiter = "\n"
    args.reverse_fast5 = args.mode == "rna"
    # One process per GPU (torchrun / torch.distributed.run): reads shard per rank -- extraction, basecalling and the
    # result files -- and are gathered on the host; no collective on the data path.
    from . import shard
    dist, rank, world, device = shard.init_distributed()
    if device is not None:
        FLAGS.device = device
    if path.isdir(FLAGS.input):
        from .extract import list_fast5, prepare_folders
        fast5_list = list_fast5(FLAGS.input, True, getattr(FLAGS, "test_number", None))
        if getattr(FLAGS, "via_signal_files", False):
            FLAGS.no_raw = False                          # the two-pass path reads raw/<name>.signal back: it must exist
        if fast5_list and not getattr(FLAGS, "via_signal_files", False):
            # Direct path: ONE partition decides which rank decodes and basecalls a fast5 file; the reader
            # threads write raw/<name>.signal for the output tree and window the decoded samples straight away -- no
            # extract-everything barrier, no text parsed back.
            import os
            from .extract import unique_read_files, logger
            prepare_folders(FLAGS, rank, world)
            fast5_list, dropped = unique_read_files(fast5_list)       # before partitioning: every rank drops the same files
            for lost, kept in (dropped if rank == 0 else []):
                logger.error("Read name of %s is taken by %s as well: the reference's extraction would overwrite the first; only the "
                             "second is basecalled." % (lost, kept))
            FLAGS.input = FLAGS.output + "/raw/"          # what the .meta files record (entry.py:38)
            sizes = {f: os.path.getsize(f) for f in fast5_list}
            FLAGS.fast5_files = shard.partition_reads(fast5_list, world, rank, sizes)
            if dist is None:
                return chiron_eval.run(args)
            out = shard.run_sharded(FLAGS, lambda fl, mine: chiron_eval.evaluation(fl, fast5_files=fl.fast5_files), dist,
                                    partition=False)
            dist.destroy_process_group()
            return out
        if fast5_list:
            # the reference's own two passes (entry.py:33-38): extract every file to raw/*.signal, then basecall raw/
            extract(FLAGS, rank, world)
            if dist is not None:
                dist.barrier()
            FLAGS.input = FLAGS.output + "/raw/"
        # else: a folder of .signal files (what extraction would have produced) is basecalled in place
    if dist is None:
        return chiron_eval.run(args)
    out = shard.run_sharded(FLAGS, lambda fl, mine: chiron_eval.evaluation(fl, file_list=mine), dist)
    dist.destroy_process_group()
    return out


def build_parser():
    parser = argparse.ArgumentParser(prog="chiron", description="A deep neural network basecaller (MI355X engine).")
    parser.add_argument("-v", "--version", action="version", version="chiron_amd version " + __version__)
    subparsers = parser.add_subparsers(title="sub command", help="sub command help")
    model_default_path = path.join(path.abspath(path.dirname(__file__)), "model", "DNA_default")
    p = subparsers.add_parser("call", description="Perform basecalling", help="Perform basecalling.")
    p.add_argument("-i", "--input", required=True, help="File path or Folder path to the fast5 file.")
    p.add_argument("-o", "--output", required=True, help="Output folder path")
    p.add_argument("-m", "--model", type=str, default=model_default_path, help="model folder path")
    p.add_argument("-s", "--start", type=int, default=None, help="Start index of the signal file.")
    p.add_argument("-b", "--batch_size", type=int, default=None, help="Batch size for run.")
    p.add_argument("-l", "--segment_len", type=int, default=None, help="Segment length to be divided into.")
    p.add_argument("-j", "--jump", type=int, default=None, help="Step size for segment")
    p.add_argument("-t", "--threads", type=int, default=None, help="Host threads, 0 = all.")
    p.add_argument("--finish-procs", type=int, default=0,
                   help="Worker processes for consensus / quality / writers (0: threads; useful behind the fp16 engine).")
    p.add_argument("-e", "--extension", default="fastq", help="Output file type.")
    p.add_argument("--beam", type=int, default=None, help="Beam width of the CTC beam search decoder, 0 = greedy.")
    p.add_argument("--concise", action="store_true", help="Only write the result files.")
    p.add_argument("--mode", default="dna", help="Output mode, dna or rna.")
    p.add_argument("--test_number", default=None, type=int, help="Extract test_number reads, default all.")
    p.add_argument("-p", "--preset", default=None, help="Preset evaluation parameters: dna-pre, rna-pre")
    p.add_argument("--device", type=int, default=0, help="HIP device ordinal.")
    p.add_argument("--gpus", type=int, default=0,
                   help="Basecall on this many GPUs of the node: the command starts one process per GPU itself (reads sharded per "
                        "process, host-side gather into merged.<ext>, per-process CPU affinity); 0 / 1: this process, --device.  "
                        "(torch.distributed.run launches are honoured as before.)")
    p.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "fp16-w2", "fp32-split"],
                   help="Engine arithmetic: fp32 (parity path), fp16 (f16 MFMA conv + LSTM, fp32 CTC), fp16-w2 (fp16's activations against "
                        "exact hi + lo weights: the f16 mode for trained checkpoints), fp32-split "
                        "(fp32 values as hi/lo half pairs on the f16 matrix cores).")
    p.add_argument("--no-raw", dest="no_raw", action="store_true",
                   help="fast5 input on the direct path: do not write raw/<name>.signal (the reference's extraction output, "
                        "extract_sig_ref.py:119-123 -- 4 bytes of text per sample that nothing reads back here).  Opt-in deviation from the "
                        "reference's output tree; ignored with --via-signal-files.")
    p.add_argument("--no-calibration", dest="no_calibration", action="store_true",
                   help="--dtype fp16: skip the bias correction for the weights' rounding to halves (Engine.calibrate on a fixed synthetic "
                        "calibration batch at start-up).")
    p.add_argument("--via-signal-files", dest="via_signal_files", action="store_true",
                   help="fast5 input: the reference's two passes (extract everything to raw/*.signal, then parse the text back) "
                        "instead of windowing the decoded samples directly; same output files.")
    p.add_argument("--synthetic-weights", dest="synthetic_weights", action="store_true",
                   help="Use seeded synthetic weights when the model folder has no checkpoint data.")
    p.set_defaults(func=evaluation)
    return parser


def main(arguments=None):
    parser = build_parser()
    argv = list(sys.argv[1:] if arguments is None else arguments)
    args = parser.parse_args(argv)
    args.child_argv = argv                    # `--gpus N` re-runs this command line in N rank processes
    if hasattr(args, "func"):
        return args.func(args)
    parser.print_help()


if __name__ == "__main__":
    main()
