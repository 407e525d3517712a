"""tinysql_amd — MI355X (gfx950) hot path for TinySQL's chunked volcano operators.

The product is `libtsq.so` (hand-written HIP kernels behind the C-ABI of include/tsq.h).  This
Python package is the harness that plays the role of the Go caller: a ctypes binding
(`_lib`), numpy-backed mirrors of util/chunk (`chunk`), an expression-tree -> bytecode compiler
mirroring package `expression` (`expression`), and `Executor` mirrors with the reference's
Open/Next/Close contract (`executor`).  There is no CPU fallback anywhere in this package:
every operator raises if libtsq.so or a GPU is missing.
"""
from . import _abi as abi  # noqa: F401

__all__ = ["abi"]
