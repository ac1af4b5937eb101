"""ctypes binding of libchiron_amd.so (include/chiron_amd.h).

The library is the product: there is no Python or CPU fallback.  Importing
this module never touches the GPU; `load()` raises if the shared object has
not been built (python -c "import __graft_entry__ as g; g.build()").
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHIRON_AMD_LIB: another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("CHIRON_AMD_LIB") or os.path.join(_HERE, "csrc", "libchiron_amd.so")

MAX_BLOCKS = 8
CLASSES = 5

ABI_VERSION = 7      # CHIRON_ABI_VERSION of include/chiron_amd.h this binding was written against
OK, ERR_INVALID, ERR_DEVICE, ERR_STATE, ERR_OVERFLOW = 0, 1, 2, 3, 4
RNN_STACK, RNN_MULTI = 0, 1
BN_POPULATION, BN_BATCH = 0, 1
F32, F16, F32_SPLIT, F16_W2 = 0, 1, 2, 3
X_ON_DEVICE, WANT_PROB, WANT_LOGITS, NO_DECODE_COPY, COMPACT_DECODE = 1, 2, 4, 8, 16
KERNAL_GLUE, KERNAL_STICK, KERNAL_SIMPLE = 1, 2, 3


class ResBlock(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("k", C.c_int32),
                ("stride", C.c_int32), ("i_bn", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("blocks", ResBlock * MAX_BLOCKS), ("rnn_kind", C.c_int32),
                ("rnn_layers", C.c_int32), ("hidden", C.c_int32), ("classes", C.c_int32),
                ("bn_mode", C.c_int32), ("stem_k", C.c_int32), ("stem_stride", C.c_int32), ("stem_channels", C.c_int32)]


class EngineOpts(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("max_batch", C.c_int32), ("segment_len", C.c_int32),
                ("n_slots", C.c_int32), ("dtype", C.c_int32), ("max_beam", C.c_int32)]


class Decoded(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("indices", C.POINTER(C.c_int64)), ("values", C.POINTER(C.c_int64)),
                ("dense_shape", C.c_int64 * 2), ("log_prob", C.POINTER(C.c_float)),
                ("prob_logits", C.POINTER(C.c_float)), ("logits", C.POINTER(C.c_float)),
                ("batch", C.c_int32), ("T", C.c_int32),
                ("flat_labels", C.POINTER(C.c_uint8)), ("row_counts", C.POINTER(C.c_int32))]


class EngineSizes(C.Structure):
    _fields_ = [("T", C.c_int32), ("ratio", C.c_double), ("largest_tensor_bytes", C.c_uint64), ("tensor_limit_bytes", C.c_uint64),
                ("slot_bytes", C.c_uint64), ("total_bytes", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_int64),
                ("flops", C.c_double), ("bytes", C.c_double)]


class PipelineOpts(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("segment_len", C.c_int32), ("jump", C.c_int32), ("start", C.c_int32), ("beam", C.c_int32),
                ("fastq", C.c_int32), ("concise", C.c_int32), ("rna", C.c_int32), ("no_raw", C.c_int32), ("n_threads", C.c_int32),
                ("n_slots", C.c_int32), ("null_engine", C.c_int32), ("null_ratio", C.c_double), ("output", C.c_char_p),
                ("delimiter", C.c_char_p), ("input_name", C.c_char_p), ("model_name", C.c_char_p), ("name_root", C.c_char_p)]


class PipelineStats(C.Structure):
    _fields_ = [("reads", C.c_int64), ("reads_finished", C.c_int64), ("windows", C.c_int64), ("batches", C.c_int64),
                ("consensus_bases", C.c_int64), ("files_failed", C.c_int64), ("seconds", C.c_double), ("messages", C.c_char * 4096)]


# every symbol include/chiron_amd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("chiron_weights_size", C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_size_t)]),
    ("chiron_engine_create", C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_size_t, C.POINTER(EngineOpts),
                                       C.POINTER(C.c_void_p)]),
    ("chiron_engine_destroy", None, [C.c_void_p]),
    ("chiron_engine_plan", C.c_int, [C.POINTER(ModelDesc), C.POINTER(EngineOpts), C.POINTER(EngineSizes)]),
    ("chiron_engine_dims", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("chiron_engine_submit", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_uint32]),
    ("chiron_engine_submit_pieces", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                              C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32]),
    ("chiron_engine_decode", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_uint32]),
    ("chiron_engine_collect", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Decoded)]),
    ("chiron_engine_sync", C.c_int, [C.c_void_p]),
    ("chiron_engine_device_results", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("chiron_engine_features", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("chiron_engine_rnn_output", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("chiron_engine_calibrate", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("chiron_engine_profile", C.c_int, [C.c_void_p, C.c_int32]),
    ("chiron_engine_profile_read", C.c_int, [C.c_void_p, C.POINTER(KernelStat), C.c_int32, C.POINTER(C.c_int32)]),
    ("chiron_parse_signal_text", C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("chiron_assemble", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("chiron_finish_read", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_char_p,
                                     C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("chiron_overlap_displacement", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                              C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("chiron_consensus_device", C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("chiron_crc32c", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("chiron_fast5_open", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("chiron_fast5_close", None, [C.c_void_p]),
    ("chiron_fast5_read_count", C.c_int32, [C.c_void_p]),
    ("chiron_fast5_read_info", C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64)]),
    ("chiron_fast5_signal", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]),
    ("chiron_fast5_fastq", C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]),
    ("chiron_write_signal_text", C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p]),
    ("chiron_pipeline_run", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int64, C.POINTER(PipelineOpts), C.POINTER(PipelineStats)]),
    ("chiron_last_error", C.c_char_p, []),
    ("chiron_device_pci_bus_id", C.c_int, [C.c_int32, C.c_char_p, C.c_size_t]),
    ("chiron_abi_version", C.c_int32, []),
    ("chiron_build_flags", C.c_uint32, []),
]
BUILD_TIMING = 1

_lib = None


class ChironError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "libchiron_amd status %d: %s" % (status, message))
        self.status = status


def _init_torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own ROCm runtime.  Two HSA runtimes cannot both open the GPU from one
    process: if libchiron_amd.so (linked against /opt/rocm) initialises HIP first, a later torch.cuda call
    fails with "No HIP GPUs are available"; the other order is fine because the dynamic loader then resolves
    our libamdhip64 dependency to the runtime torch already loaded.  So when torch is already imported (the
    caller hands us device tensors, bench.py, torchrun sharding) make sure its runtime is up before dlopen."""
    torch = sys.modules.get("torch")
    if torch is None:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # a torch build without GPU support is irrelevant to the engine
        pass


def load():
    """Load the HIP library.  Fails loudly when it is missing: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). chiron_amd has no CPU fallback." % LIB_PATH)
    _init_torch_runtime_first()
    lib = C.CDLL(LIB_PATH)
    # the version first: a stale library lacks the newer symbols, and "rebuild it" is the message to give, not an AttributeError
    try:
        lib.chiron_abi_version.restype = C.c_int32
        have = lib.chiron_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise ImportError("%s has ABI version %s, this binding was written against %d: rebuild it "
                          "(python -c 'import __graft_entry__ as g; g.build()')" % (LIB_PATH, have, ABI_VERSION))
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if (lib.chiron_build_flags() & BUILD_TIMING) and os.environ.get("CHIRON_ALLOW_TIMING_BUILD") != "1":
        raise ImportError("%s is a TIMING build (a kernel variant with parts switched off, csrc/timing_variants.h): its results "
                          "are garbage.  Measurement tools set CHIRON_ALLOW_TIMING_BUILD=1; nothing else may load it." % LIB_PATH)
    _lib = lib
    return lib


def check(status):
    if status != OK:
        raise ChironError(status, load().chiron_last_error().decode("utf-8", "replace"))
