"""GPU parity: fused expression kernels (through the C-ABI) vs the oracle, on the reference's own
random generators (expression/bench_test.go:56-152).  Integer results bit-exact; real results
bit-exact as well (same IEEE operations in the same order, -ffp-contract=off)."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column
from tinysql_amd.executor import MockDataSource, ProjectionExec, SelectionExec, drain

from . import helpers as H
from .test_hostsim_vs_oracle import all_exprs, cols_for

pytestmark = pytest.mark.gpu


def check_same(ctx, orc, e, chk, jit=None):
    prog = E.compile_expr(e)
    ce = E.CompiledExpr(ctx, [e], jit=jit)
    try:
        try:
            want, ow = orc.expr_eval(prog, chk)
        except orc.OracleError as err:
            with pytest.raises(_lib.TsqError) as ei:
                ce.VecEval(chk)
            assert ei.value.status == err.status
            return
        got = ce.VecEval(chk)
        assert ce.warnings == ow
        gn = got.notnull if got.notnull is not None else np.ones(len(got), bool)
        wn = want.notnull if want.notnull is not None else np.ones(len(want), bool)
        assert (gn == wn).all()
        assert (got.data.view(np.uint64)[gn] == want.data.view(np.uint64)[wn]).all()
        if jit == abi.JIT_FORCE:
            assert ce.jit_launches() >= 1  # the specialised kernel really served the call
    finally:
        ce.close()


@pytest.mark.parametrize("small", [True, False], ids=["small-ints", "full-range"])
def test_every_signature_matches_oracle(ctx, orc, small):
    rng = np.random.default_rng(31 if small else 32)
    chk = cols_for(rng, 1024, small)
    for e in all_exprs():
        check_same(ctx, orc, e, chk)


def test_selection_vector_and_big_batch(ctx, orc):
    rng = np.random.default_rng(33)
    chk = cols_for(rng, 200_000, True)  # tidb_max_chunk_size raised: one PCIe round trip for 200k rows
    for e in all_exprs()[:12]:
        check_same(ctx, orc, e, chk)
    chk = cols_for(rng, 1024, True)
    chk.sel = np.sort(rng.choice(1024, 333, replace=False)).astype(np.int32)
    for e in all_exprs():
        check_same(ctx, orc, e, chk)


def test_filter_vs_vec_eval_bool(ctx, orc):
    rng = np.random.default_rng(34)
    chk = cols_for(rng, 5000, True)
    I, R = abi.I64, abi.F64
    c = {i: E.Column(i, t) for i, t in enumerate([I, I, abi.U64, abi.U64, R, R, abi.F32])}
    F = E.ScalarFunction
    lists = [[F("gt", c[0], E.Constant(0))], [c[0], c[4]],
             [F("gt", c[0], E.Constant(-50000)), F("lt", F("plus", c[0], c[1]), E.Constant(1000)), c[5]],
             [F("or", F("isnull", c[0]), F("gt", c[1], E.Constant(0))), F("div", c[4], F("minus", c[5], c[5]))]]
    for lst in lists:
        ce = E.CompiledExpr(ctx, lst)
        try:
            sel, nulls = ce.VectorizedFilter(chk, want_nulls=True)
            osel, onull, ow = orc.filter_eval(ce.progs, len(lst), chk)
            assert (sel == osel).all() and (nulls == onull).all() and ce.warnings == ow
        finally:
            ce.close()


def test_first_error_node_then_row(ctx, orc):
    i64max, u64max = (1 << 63) - 1, (1 << 64) - 1
    chk = H.chunk_from_rows([[i64max, 1, 2], [1, u64max, 1]], [abi.I64, abi.U64, abi.U64])
    F = E.ScalarFunction
    e = F("mul", F("mul", E.Column(0, abi.I64), F("in", E.Column(0, abi.I64), E.Constant(i64max))),
          F("isnull", F("plus", E.Column(1, abi.U64), E.Column(2, abi.U64))))
    check_same(ctx, orc, e, chk)


def test_selection_and_projection_executors(ctx, orc):
    # Q3-style: WHERE o_orderdate < D AND l_shipdate > D ; SELECT price * (1 - discount)
    rng = np.random.default_rng(35)
    n = 20000
    chk = Chunk([Column(abi.I64, rng.integers(9000, 9500, n)), Column(abi.I64, rng.integers(9000, 9500, n)),
                 Column(abi.F64, rng.random(n) * 1e5), Column(abi.F64, rng.integers(0, 11, n) / 100.0, rng.random(n) > 0.05)])
    t = [abi.I64, abi.I64, abi.F64, abi.F64]
    F = E.ScalarFunction
    filt = [F("lt", E.Column(0, abi.I64), E.Constant(9250)), F("gt", E.Column(1, abi.I64), E.Constant(9250))]
    proj = [F("mul", E.Column(2, abi.F64), F("minus", E.Constant(1.0), E.Column(3, abi.F64))), E.Column(0, abi.I64)]
    exe = ProjectionExec(ctx, SelectionExec(ctx, MockDataSource(ctx, chk, 4096), filt, 4096), proj, 4096)
    rows = [r for c in drain(exe) for r in c.rows()]
    sel, _, _ = orc.filter_eval(E.compile_list(filt), 2, chk)
    rev, _ = orc.expr_eval(E.compile_expr(proj[0]), chk)
    want = [(rev.values()[i], int(chk.columns[0].data[i])) for i in np.nonzero(sel)[0]]
    assert rows == want  # Projection/Selection preserve child order (projection.go:187-207)


def test_jit_specialised_kernels_match_oracle(ctx, orc):
    # every builtin signature once more, through hiprtc-specialised kernels (same tsq_eval_row source with the program
    # as a compile-time constant): results, NULLs, warnings and the first-error semantics must not change
    rng = np.random.default_rng(36)
    small, full = cols_for(rng, 3000, True), cols_for(rng, 3000, False)
    exprs = all_exprs()
    for i, e in enumerate(exprs):
        check_same(ctx, orc, e, small if i % 2 == 0 else full, jit=abi.JIT_FORCE)
    i64max, u64max = (1 << 63) - 1, (1 << 64) - 1
    chk = H.chunk_from_rows([[i64max, 1, 2], [1, u64max, 1]], [abi.I64, abi.U64, abi.U64])
    F = E.ScalarFunction
    e = F("mul", F("mul", E.Column(0, abi.I64), F("in", E.Column(0, abi.I64), E.Constant(i64max))),
          F("isnull", F("plus", E.Column(1, abi.U64), E.Column(2, abi.U64))))
    check_same(ctx, orc, e, chk, jit=abi.JIT_FORCE)
    # CNF filters
    chk = cols_for(rng, 5000, True)
    I, R = abi.I64, abi.F64
    c = {i: E.Column(i, t) for i, t in enumerate([I, I, abi.U64, abi.U64, R, R, abi.F32])}
    lists = [[F("gt", c[0], E.Constant(0))], [c[0], c[4]],
             [F("gt", c[0], E.Constant(-50000)), F("lt", F("plus", c[0], c[1]), E.Constant(1000)), c[5]],
             [F("or", F("isnull", c[0]), F("gt", c[1], E.Constant(0))), F("div", c[4], F("minus", c[5], c[5]))]]
    for lst in lists:
        ce = E.CompiledExpr(ctx, lst, jit=abi.JIT_FORCE)
        try:
            sel, nulls = ce.VectorizedFilter(chk, want_nulls=True)
            osel, onull, ow = orc.filter_eval(ce.progs, len(lst), chk)
            assert (sel == osel).all() and (nulls == onull).all() and ce.warnings == ow
            assert ce.jit_launches() >= 1
        finally:
            ce.close()


@pytest.mark.parametrize("n", [4096, 10_007, 131_073])
def test_jit_four_rows_per_lane_and_the_bitmap_written_by_the_kernel(ctx, orc, n):
    # round 6: the specialised kernel evaluates four rows per lane from 16-byte loads and, for batches of 4096 rows or more, writes whole
    # 32-row words of the result's null bitmap itself (tsq_expr.hip: jit_expr, counters[2]); the rows past the last whole word come
    # through the flag bytes and the pack pass.  NULL inputs, NULL results, every signature, and an overflow in the middle of a batch
    # (the reference's first-error row, builtin_arithmetic_vec.go:389-420) at sizes around the word boundaries
    rng = np.random.default_rng(n)
    small, full = cols_for(rng, n, True), cols_for(rng, n, False)
    for i, e in enumerate(all_exprs()):
        check_same(ctx, orc, e, small if i % 2 == 0 else full, jit=abi.JIT_FORCE)
    a = rng.integers(-1000, 1000, n)
    a[n // 2 + 3] = (1 << 63) - 1
    from tinysql_amd.chunk import Chunk, Column
    chk = Chunk([Column(abi.I64, a, rng.random(n) > 0.1), Column(abi.I64, np.ones(n, np.int64))])
    e = E.ScalarFunction("plus", E.Column(0, abi.I64), E.Column(1, abi.I64))
    check_same(ctx, orc, e, chk, jit=abi.JIT_FORCE)  # row n/2 + 3 overflows (unless it is NULL: then the batch passes) — same status either way


@pytest.mark.parametrize("variant", [0, 4, 3, 7, 12, 15, 8 | 16, 4 | 64])
@pytest.mark.parametrize("n", [4096, 10_007, 131_073])
def test_jit_variants_agree_with_the_oracle(ctx, orc, n, variant):
    # TSQ_KNOB_JIT_VARIANT: the A/B forms of jit_expr (non-temporal accesses, the whole-wave coalesced row layout with its own bitmap words,
    # two steps in flight, other grid sizes) are the same function of the rows — every signature, NULLs in and out, an overflow mid-batch
    ctx.set_knob(abi.KNOB_JIT_VARIANT, variant)
    try:
        rng = np.random.default_rng(n + variant)
        small, full = cols_for(rng, n, True), cols_for(rng, n, False)
        for i, e in enumerate(all_exprs()):
            check_same(ctx, orc, e, small if i % 2 == 0 else full, jit=abi.JIT_FORCE)
        a = rng.integers(-1000, 1000, n)
        a[n // 2 + 3] = (1 << 63) - 1
        from tinysql_amd.chunk import Chunk, Column
        chk = Chunk([Column(abi.I64, a, rng.random(n) > 0.1), Column(abi.I64, np.ones(n, np.int64))])
        check_same(ctx, orc, E.ScalarFunction("plus", E.Column(0, abi.I64), E.Column(1, abi.I64)), chk, jit=abi.JIT_FORCE)
    finally:
        ctx.set_knob(abi.KNOB_JIT_VARIANT, -1)


def test_jit_auto_compiles_on_a_helper_thread_and_never_blocks(ctx, orc):
    # TSQ_JIT_AUTO (round 6): the hiprtc compile of a plan runs on a helper thread once the handle has seen 256 Ki rows; the calls made
    # meanwhile are served by the interpreter kernels (same rows), and once the code object is there the specialised kernel takes over
    import time
    F = E.ScalarFunction
    rng = np.random.default_rng(77)
    n = 300_000
    chk = cols_for(rng, n, True)
    e = F("minus", F("mul", F("plus", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.Constant(7)), E.Column(0, abi.I64))
    ce = E.CompiledExpr(ctx, [e])  # (no jit argument: AUTO)
    try:
        want, _ = orc.expr_eval(E.compile_expr(e), chk)
        wn = want.notnull if want.notnull is not None else np.ones(len(want), bool)
        t0 = time.time()
        first_call = None
        calls = 0
        while time.time() - t0 < 60:
            t = time.time()
            got = ce.VecEval(chk)
            if first_call is None:
                first_call = time.time() - t
            calls += 1
            gn = got.notnull if got.notnull is not None else np.ones(len(got), bool)
            assert (gn == wn).all() and (got.data.view(np.uint64)[gn] == want.data.view(np.uint64)[wn]).all()
            if ce.jit_launches() >= 1:
                break
            time.sleep(0.01)
        assert ce.jit_launches() >= 1, "the helper thread's module never arrived"
        assert calls >= 2  # the first call could not have waited for the compile ...
        assert first_call < 0.15  # ... and did not (hiprtc takes ~250 ms)
        assert ce.jit_compile_ms() > 1.0
    finally:
        ce.close()


def test_a_process_may_end_while_the_helper_thread_still_compiles():
    # a short script: one AUTO evaluation of 300 000 rows starts the background hiprtc compile, then the process ends without closing
    # anything — the library's atexit hook waits for the compile (hiprtc's teardown under a running compile would crash the exit)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np\n"
        "from tinysql_amd import _abi as abi, _lib, expression as E\n"
        "from tinysql_amd.chunk import Chunk, Column\n"
        "ctx = _lib.Context(0)\n"
        "n = 300000\n"
        "chk = Chunk([Column(abi.I64, np.arange(n)), Column(abi.I64, np.arange(n))])\n"
        "ce = E.CompiledExpr(ctx, [E.ScalarFunction('plus', E.Column(0, abi.I64), E.Column(1, abi.I64))])\n"
        "got = ce.VecEval(chk)\n"
        "assert (got.data == 2 * np.arange(n)).all()\n"
        "print('evaluated', flush=True)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "evaluated" in r.stdout, (r.returncode, r.stdout[-200:], r.stderr[-400:])
