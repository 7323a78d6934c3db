"""cudf_amd -- MI355X-native (gfx950) implementation of the cudf hot path:
sort / sorted_order / hash join / groupby / reduce / scan over Arrow-layout column buffers.

Importing the package loads cudf_amd/libcudf_amd.so (hand-written HIP kernels behind the C ABI of
include/cudf_amd/gx.h).  There is NO CPU fallback: a missing library is an ImportError.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from .column import Column  # noqa: F401
from . import ops  # noqa: F401
from .dataframe import DataFrame  # noqa: F401

__version__ = "0.1.0"
