"""chiron_amd -- MI355X-native basecalling inference engine behind the
`chiron call` / chiron_eval hot path of haotianteng/Chiron.

The compute path is libchiron_amd.so (hand-written HIP for gfx950, C ABI in
include/chiron_amd.h).  There is no CPU fallback: importing the package is
cheap and GPU-free, but creating an Engine without the built library or
without a GPU raises.
"""
__version__ = "0.1.0"

from .model import (ModelSpec, dna_default_spec, rna_default_spec, rna_head_spec, synthetic_weights,  # noqa: F401
                    synthetic_signal, read_config, spec_from_variables, spec_from_config, load_model)
from .engine import Engine, SparseTensor, DecodeResult, seq_len_for_engine  # noqa: F401
