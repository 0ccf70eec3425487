"""cutesv_amd — MI355X-native clustering-and-refinement stage of cuteSV.

Only the hot path lives here: flat signature columns in, the reference's row
lists out, hand-written HIP kernels in between (cutesv_amd/csrc, C ABI in include/cutesv_hip.h).
"""
from .columns import SigStore, Params, NameTable, TYPES  # noqa: F401

__version__ = "0.1.0"
