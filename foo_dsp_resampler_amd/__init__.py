"""foo_dsp_resampler_amd -- Python binding of the MI355X SoX-rate engine.

The product is the C-ABI shared library `libratelib_amd.so` (include/ratelib.h, include/ratelib_amd.h)
whose entry points are what the reference plugin binds (rate/ratelib.h:72-81).  This package is a thin
ctypes mirror of that interface used by the tests and bench.py; it contains no signal processing and
no CPU fallback: if the library (or a GPU) is missing, calls fail.
"""
from .ratelib import (RRConfig, RRError, Resampler, available_symbols, describe_dispatch, describe_plan, lib, lib_path,  # noqa: F401
                      plan_table, RR_BEST, RR_NORM, EXPECTED_SYMBOLS)

__all__ = ["RRConfig", "RRError", "Resampler", "describe_plan", "describe_dispatch", "plan_table", "lib", "lib_path",
           "available_symbols", "RR_BEST", "RR_NORM", "EXPECTED_SYMBOLS"]
