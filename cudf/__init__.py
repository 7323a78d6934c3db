"""`import cudf` -- the name a user of the reference already types (python/cudf/cudf/__init__.py).

An alias package over :mod:`cudf_amd`, nothing more: `cudf.DataFrame` IS `cudf_amd.DataFrame` (the thin wrapper of
python/cudf/cudf/core/dataframe.py's `sort_values` / `merge` / `groupby(...).agg` that SURVEY.md section 2 scopes in), `cudf.from_pandas`
mirrors the reference's module-level constructor (python/cudf/cudf/core/dataframe.py `from_pandas`).  Importing it loads
cudf_amd/libcudf_amd.so exactly as `import cudf_amd` does -- there is no CPU fallback behind this name either.  Everything else of
the reference's Python package (Series arithmetic, strings, I/O, indexes ...) is out of scope and NOT here: an attribute that is not
listed in ``__all__`` raises AttributeError instead of pretending.
"""
import cudf_amd as _impl
from cudf_amd import DataFrame  # noqa: F401

__version__ = _impl.__version__
__all__ = ["DataFrame", "from_pandas", "__version__"]


def from_pandas(obj):
    """cudf.from_pandas(pandas.DataFrame) -> cudf.DataFrame (device-resident columns)."""
    import pandas as pd
    if not isinstance(obj, pd.DataFrame):
        raise TypeError("cudf.from_pandas: only pandas.DataFrame is supported on this path")
    return DataFrame.from_pandas(obj)


def __getattr__(name):
    raise AttributeError(f"module 'cudf' (MI355X hot-path build) has no attribute {name!r}: only {__all__} are provided; "
                         "see cudf_amd for the column-level operators")
