"""CPU: the C-ABI library loads, exports every symbol include/tsq.h declares, and fails LOUDLY
(no CPU fallback) when no GPU is visible."""
import ctypes as C
import os
import re

import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tsq.h")).read()
    return sorted(set(re.findall(r"\b(tsq_[a-z0-9_]+)\s*\(", txt)) - {"tsq_status"})


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(abi.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), "libtsq.so does not export %s" % name
    assert lib.tsq_abi_version() == abi.TSQ_ABI_VERSION


def test_struct_layouts_match_the_header():
    # sizes the C compiler produces for include/tsq.h (asserted again on the C side by the build)
    assert C.sizeof(abi.Col) == 48
    assert C.sizeof(abi.ExprOp) == 8
    assert C.sizeof(abi.ExprProg) == 16 + 8 * abi.EXPR_MAX_OPS + 8 * abi.EXPR_MAX_CONSTS + 8 + abi.EXPR_STR_POOL
    assert C.sizeof(abi.GenSpec) == 56
    assert C.sizeof(abi.AggFunc) == 24


def test_struct_sizes_equal_what_gcc_computes_for_the_header(tmp_path):
    # every struct that crosses the boundary by pointer: compile include/tsq.h with gcc and compare sizeof with ctypes
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(tsq_col), sizeof(tsq_expr_prog), '
                   'sizeof(tsq_gen_spec), sizeof(tsq_join_cfg), sizeof(tsq_agg_cfg), sizeof(tsq_sort_cfg), sizeof(tsq_stats), sizeof(tsq_rowcodec_col));return 0;}\n'
                   % os.path.join(ROOT, "include", "tsq.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(t) for t in (abi.Col, abi.ExprProg, abi.GenSpec, abi.JoinCfg, abi.AggCfg, abi.SortCfg, abi.Stats, abi.RowcodecCol)]


def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.load()
    if lib.tsq_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    st = lib.tsq_ctx_create(0, C.byref(h))
    assert st == abi.ERR_NO_DEVICE and not h.value
    assert "no CPU fallback" in _lib.last_error()
    with pytest.raises(_lib.TsqError):
        _lib.Context(0)


def test_product_package_never_imports_the_oracle():
    # the oracle is the checker, never the product: no import / include / dlopen of anything under oracle/
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+\s*oracle|liboracle|oracle/|oracle\.h|orc_[a-z_]+\s*\()")
    pkg = os.path.join(ROOT, "tinysql_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".go", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                m = bad.search(src)
                assert not m, "%s references the oracle (%r): the product path must not depend on it" % (f, m.group(0))
