"""gtars_amd -- MI355X-native engine for the gtars interval-overlap /
region-set tokenization hot path (gtars-overlaprs, gtars-tokenizers, gtars-igd).

The compute path is libgtars_amd.so (hand-written HIP for gfx950 behind the C
ABI in include/gtars_amd.h).  Importing this package loads that library and
fails loudly if it has not been built; there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from ._lib import KIND_AILIST, KIND_BITS, UNKNOWN_CHROM, CapacityError, GtarsError, NoDeviceError, device_count, reload_env
from .engine import IgdIndex, OverlapIndex

__all__ = [
    "OverlapIndex",
    "IgdIndex",
    "KIND_BITS",
    "KIND_AILIST",
    "UNKNOWN_CHROM",
    "GtarsError",
    "NoDeviceError",
    "CapacityError",
    "device_count",
    "reload_env",
]
__version__ = "0.1.0"
