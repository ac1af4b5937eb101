"""fast5 -> <out>/raw/<name>.signal (counterpart of chiron/utils/extract_sig_ref.py).

Uses the package's own minimal HDF5 reader (fast5.py); h5py is not required.  Unreadable reads are
logged to <out>/log/extract.log and skipped (extract_sig_ref.py:97-117)."""
import logging
import os
from multiprocessing import Pool, cpu_count

import numpy as np

logger = logging.getLogger(name="chiron_call")


def set_logger(log_file):
    hd = logging.FileHandler(log_file)
    hd.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    logger.addHandler(hd)
    logger.propagate = False
    logger.setLevel(logging.INFO)


def extract_file(path, mode="dna", unit=False):
    """extract_sig_ref.py:149-175 / :178-193 -> list of (suffix, raw_signal, reference, read_id)."""
    from . import fast5
    out = []
    for rec in fast5.read_fast5(path):
        raw = np.asarray(rec["signal"])
        if unit and rec.get("channel") is not None:
            ch = rec["channel"]            # (raw+offset)*range/digitisation, extract_sig_ref.py:153-158
            raw = (raw + float(ch["offset"])) * float(ch["range"]) / float(ch["digitisation"])
        if mode == "rna":
            raw = raw[::-1]                # extract_sig_ref.py:165
        out.append((rec.get("suffix", ""), raw, rec.get("fastq", ""), rec.get("read_id", "")))
    return out


def extract_records(full, FLAGS):
    """extract_file_wrapper (extract_sig_ref.py:92-147) for ONE fast5 file: every read's raw signal is written to
    raw/<name>.signal (and its reference to reference/<stem>_ref.fastq) exactly as the reference's extraction does, and
    returned as [(name, float32 signal)] so that the caller can window it directly instead of parsing the text back
    (SURVEY 8(f)1).  The hot path (`chiron call`: unit = False, entry.py:36) reads through the native reader and writes
    through the native text writer (csrc/fast5.cpp); the pA conversion (unit = True) keeps the Python reader, which also
    decodes the channel attributes.  Unreadable files are logged and skipped (:97-117): -> []."""
    from . import fast5
    stem = os.path.splitext(os.path.basename(full))[0]
    unit = getattr(FLAGS, "unit", False)
    try:
        if unit:
            recs = [(s, np.asarray(r), ref, rid) for s, r, ref, rid in extract_file(full, FLAGS.mode, True)]
        else:
            recs = [(r["suffix"], r["signal"], r["fastq"], r["read_id"])
                    for r in fast5.read_fast5_native(full, reverse=FLAGS.mode == "rna")]     # extract_sig_ref.py:165
        if not recs:
            raise ValueError("Fail in extracting raw signal.")
    except Exception as e:  # noqa: BLE001 -- skip-and-log like the reference
        logger.error("Cannot extract file %s. %s" % (full, e))
        return []
    out = []
    for suffix, raw, reference, read_id in recs:
        if len(raw) == 0:
            logger.error("Cannot extract file %s. Got empty raw signal" % full)
            continue
        name = read_id if getattr(FLAGS, "idname", False) else stem + suffix
        sig_path = os.path.join(FLAGS.raw_folder, name + ".signal")
        if getattr(FLAGS, "no_raw", False):
            pass        # `chiron call --no-raw`: the direct path never reads raw/<name>.signal back; skipping it is an opt-in deviation
        elif unit:
            with open(sig_path, "w+") as f:
                f.write(FLAGS.delimiter.join([str(v) for v in raw.tolist()]))     # extract_sig_ref.py:122-123
        else:
            fast5.write_signal_text(sig_path, raw, FLAGS.delimiter)
        if len(reference) > 0:
            head = "@%s\n" % stem
            with open(os.path.join(FLAGS.ref_folder, stem + "_ref.fastq"), "w+") as f:
                f.write(head + "\n".join(reference.split("\n")[1:]))
        out.append((name, np.asarray(raw, dtype=np.float32)))
    return out


def _wrapper(args):
    full, FLAGS = args
    return len(extract_records(full, FLAGS))


def list_fast5(root_folder, recursive=True, test_number=None):
    """Every *.fast5 under root_folder, sorted by path (the reference walks in os.walk order, extract_sig_ref.py:62-79;
    sorted here so that every rank of a sharded run sees the same list)."""
    files = []
    for dirpath, _, filenames in os.walk(root_folder):
        files += [os.path.join(dirpath, fn) for fn in filenames if fn.endswith("fast5")]
        if not recursive:
            break
    files.sort()
    return files[:test_number] if test_number else files


def unique_read_files(files):
    """Two fast5 files with the same basename in different sub-folders give ONE raw/<stem>.signal in the reference (the walk is
    recursive, the output name is the stem: extract_sig_ref.py:62-79, :119): the later file overwrites the earlier and one read is
    basecalled.  The direct path keys a read by that name as well, so both files would be windowed into one record -- mixed, or
    never finished, and on different ranks they would race on the same output files.  Keep the LAST file of every stem (list_fast5
    order, what overwriting leaves behind); -> (files kept, [(dropped, kept)])."""
    last = {}
    for f in files:
        last[os.path.splitext(os.path.basename(f))[0]] = f
    keep = [f for f in files if last[os.path.splitext(os.path.basename(f))[0]] == f]
    dropped = [(f, last[os.path.splitext(os.path.basename(f))[0]]) for f in files if last[os.path.splitext(os.path.basename(f))[0]] != f]
    return keep, dropped


def prepare_folders(FLAGS, rank=0, world=1):
    """extract_sig_ref.py:43-56: raw/ reference/ log/ under the output folder, and the extraction log."""
    root_folder, out_folder = FLAGS.input_dir, FLAGS.output_dir
    if not os.path.isdir(root_folder):
        raise IOError("Input directory does not found.")
    os.makedirs(out_folder, exist_ok=True)
    FLAGS.raw_folder = os.path.abspath(os.path.join(out_folder, "raw"))
    FLAGS.ref_folder = os.path.abspath(os.path.join(out_folder, "reference"))
    FLAGS.log_folder = os.path.abspath(os.path.join(out_folder, "log"))
    for d in (FLAGS.raw_folder, FLAGS.ref_folder, FLAGS.log_folder):
        os.makedirs(d, exist_ok=True)
    set_logger(os.path.join(FLAGS.log_folder, "extract.log" if world == 1 else "extract.rank%d.log" % rank))


def extract(FLAGS, rank=0, world=1):
    """extract_sig_ref.py:31-90.  In a sharded run (one process per GPU) every rank extracts its own share of the
    file list, file k -> rank k mod world, with its own worker pool -- the counterpart of the reference's
    Pool(cpu_count()) (:58-60, :81) spread over the ranks instead of serialised on rank 0."""
    root_folder = FLAGS.input_dir
    prepare_folders(FLAGS, rank, world)
    threads = FLAGS.threads if getattr(FLAGS, "threads", 0) else max(1, cpu_count() // max(world, 1))
    files = list_fast5(root_folder, getattr(FLAGS, "recursive", True), getattr(FLAGS, "test_number", None))[rank::max(world, 1)]
    if threads > 1 and len(files) > 1:
        with Pool(min(threads, len(files))) as pool:
            counts = pool.map(_wrapper, [(f, FLAGS) for f in files])
    else:
        counts = [_wrapper((f, FLAGS)) for f in files]
    return int(sum(counts))
