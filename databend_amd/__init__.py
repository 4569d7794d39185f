"""databend_amd — MI355X-native kernels for Databend's column-batch execution hot path.

The product is ``libdbhip.so`` (hand-written HIP for gfx950 behind the C-ABI in
``include/dbhip.h``).  This package is the thin Python host used by the tests and
``bench.py``: it loads the library with ctypes and mirrors the reference's column
vocabulary (Column / DataBlock / selection vectors).  There is no CPU fallback:
every entry point raises when the HIP library or a GPU is missing.
"""
from ._lib import DbhipError, lib, load_library, library_path  # noqa: F401
