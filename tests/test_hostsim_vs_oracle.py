"""CPU differential tests: the per-row device semantics (tinysql_amd/csrc/tsq_device.h, compiled
with g++ into tests/hostsim/hostsim.so) against the oracle, on the reference's own random
generators (expression/bench_test.go:56-152: 20% NULL, +-Int63 / +-1e6*U(0,1), 1024-row chunks).

This is how the expression interpreter, the error ordering, the generator and the hashes are
verified on the CPU box; the `-m gpu` tests then check that the HIP kernels reproduce the same
results through the C-ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column, make_cols

from . import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_expr_eval.restype = C.c_int32
    lib.sim_expr_eval.argtypes = [C.POINTER(abi.ExprProg), C.POINTER(abi.Col), C.c_int32, C.c_int64, P, P, P, C.POINTER(C.c_int64)]
    lib.sim_filter_eval.restype = C.c_int32
    lib.sim_filter_eval.argtypes = [C.POINTER(abi.ExprProg), C.c_int32, C.POINTER(abi.Col), C.c_int32, C.c_int64, P, P, P,
                                    C.POINTER(C.c_int64)]
    lib.sim_gen_column.restype = None
    lib.sim_gen_column.argtypes = [C.POINTER(abi.GenSpec), C.c_int64, P, P, P]
    lib.sim_validate.restype = C.c_int32
    lib.sim_validate.argtypes = [C.POINTER(abi.ExprProg), C.c_int32]
    lib.sim_key_rank.restype = C.c_uint32
    lib.sim_key_rank.argtypes = [C.c_uint64, C.c_uint32]
    lib.sim_mulhi64.restype = C.c_uint64
    lib.sim_mulhi64.argtypes = [C.c_uint64, C.c_uint64]
    lib.sim_rowhash.restype = C.c_uint64
    lib.sim_rowhash.argtypes = [P, P, C.c_int32]
    return lib


def sim_eval(sim, prog, chk):
    keep = []
    cols = make_cols(chk.columns, keep)
    n = chk.NumRows()
    out = np.zeros(max(n, 1), np.uint64)
    nn = np.zeros(max(n, 1), np.uint8)
    w = C.c_int64(0)
    sel = chk.sel.ctypes.data_as(C.c_void_p) if chk.sel is not None else None
    st = sim.sim_expr_eval(C.byref(prog), cols, len(chk.columns), n, sel, out.ctypes.data_as(C.c_void_p), nn.ctypes.data_as(C.c_void_p),
                           C.byref(w))
    return st, out[:n], nn[:n].astype(bool), w.value


def check_same(sim, orc, e, chk):
    prog = E.compile_expr(e)
    assert sim.sim_validate(C.byref(prog), len(chk.columns)) == abi.OK
    st, out, nn, w = sim_eval(sim, prog, chk)
    try:
        col, ow = orc.expr_eval(prog, chk)
    except orc.OracleError as err:
        assert st == err.status, "device interpreter must report the same (first) error as the node-at-a-time evaluator"
        return
    assert st == abi.OK
    assert w == ow
    onn = col.notnull if col.notnull is not None else np.ones(len(col), bool)
    assert (nn == onn).all()
    a = out[nn]
    b = col.data.view(np.uint64)[onn] if col.data.dtype != np.uint64 else col.data[onn]
    assert (a == b).all(), "bit-exact (ints) / same IEEE op order (reals)"


I, U, R = abi.I64, abi.U64, abi.F64


def cols_for(rng, n, small=False):
    kw = dict(lo=-100000, hi=100000) if small else {}
    return Chunk([H.random_column(rng, I, n, **kw), H.random_column(rng, I, n, **kw),
                  H.random_column(rng, U, n, lo=0 if small else None, hi=100000 if small else None),
                  H.random_column(rng, U, n, lo=0 if small else None, hi=100000 if small else None),
                  H.random_column(rng, R, n), H.random_column(rng, R, n),
                  Column(abi.F32, (rng.random(n) * 100).astype(np.float32), rng.random(n) > 0.2)])


def all_exprs():
    c = {i: E.Column(i, t) for i, t in enumerate([I, I, U, U, R, R, abi.F32])}
    F = E.ScalarFunction
    ex = []
    for name in ("plus", "minus", "mul"):
        ex += [F(name, c[0], c[1]), F(name, c[2], c[3]), F(name, c[0], c[2]), F(name, c[2], c[0]), F(name, c[4], c[5])]
    ex += [F("minus", c[2], c[3], no_unsigned_subtraction=True), F("minus", c[0], c[3], no_unsigned_subtraction=True),
           F("minus", c[2], c[1], no_unsigned_subtraction=True)]
    ex += [F("div", c[4], c[5]), F("div", c[4], F("minus", c[5], c[5])), F("plus", c[4], c[6]), F("mul", c[6], c[6])]
    for name in ("lt", "le", "gt", "ge", "eq", "ne"):
        ex += [F(name, c[0], c[1]), F(name, c[2], c[3]), F(name, c[0], c[2]), F(name, c[2], c[0]), F(name, c[4], c[5])]
    b0, b1 = F("gt", c[0], E.Constant(0)), F("lt", c[1], E.Constant(0))
    ex += [F("and", b0, b1), F("or", b0, b1), F("and", c[0], c[1]), F("or", c[0], c[1]), F("not", c[0]), F("not", c[4]),
           F("not", b0), F("unaryminus", c[0]), F("unaryminus", c[4]), F("isnull", c[0]), F("isnull", c[4]),
           F("ifnull", c[0], c[1]), F("ifnull", c[4], c[5]), F("ifnull", c[0], E.Constant(7)),
           F("if", b0, c[0], c[1]), F("if", c[0], c[4], c[5]), F("if", F("isnull", c[0]), E.Constant(1.5), c[4]),
           F("in", c[0], c[1], E.Constant(3), E.Constant(None, E.ETInt)), F("in", c[0], c[2], c[1]), F("in", c[2], c[0], c[3]),
           F("in", c[4], c[5], E.Constant(2.5)), F("in", c[0], E.Constant(5), E.Constant(-7)),
           # a deeper tree (Q3-style revenue): price * (1 - discount) + tax / qty
           F("plus", F("mul", c[4], F("minus", E.Constant(1.0), c[5])), F("div", c[4], c[5]))]
    return ex


@pytest.mark.parametrize("small", [True, False], ids=["small-ints", "full-range"])
def test_every_signature_matches_oracle(sim, orc, small):
    rng = np.random.default_rng(1 if small else 2)
    chk = cols_for(rng, 1024, small)
    for e in all_exprs():
        check_same(sim, orc, e, chk)


def test_with_selection_vector(sim, orc):
    rng = np.random.default_rng(3)
    chk = cols_for(rng, 1024, True)
    chk.sel = np.sort(rng.choice(1024, 300, replace=False)).astype(np.int32)  # bench_test.go:650 random Sel
    for e in all_exprs()[:20]:
        check_same(sim, orc, e, chk)


def test_edge_values(sim, orc):
    i64max, i64min, u64max = (1 << 63) - 1, -(1 << 63), (1 << 64) - 1
    iv = [0, 1, -1, i64max, i64min, i64max - 1, i64min + 1, 2, -2, 1 << 62, -(1 << 62), None]
    uv = [0, 1, u64max, u64max - 1, 1 << 63, (1 << 63) - 1, (1 << 63) + 1, 2, 3, 1 << 62, 5, None]
    fv = [0.0, -0.0, 1.0, -1.0, 1.7976931348623157e308, -1.7976931348623157e308, 5e-324, float("inf"), float("nan"), 0.4, 0.5, None]
    F = E.ScalarFunction
    for a in range(len(iv)):
        rows = [[iv[a], iv[b], uv[a], uv[b], fv[a], fv[b], 1.0] for b in range(len(iv))]
        chk = H.chunk_from_rows(rows, [I, I, U, U, R, R, abi.F32])
        # errors abort a whole chunk; evaluate row by row so every pair is actually compared
        for r in range(len(rows)):
            one = chk.slice(r, r + 1)
            for e in all_exprs():
                check_same(sim, orc, e, one)


def test_filter_rows_match_vec_eval_bool(sim, orc):
    rng = np.random.default_rng(5)
    chk = cols_for(rng, 1024, True)
    c = {i: E.Column(i, t) for i, t in enumerate([I, I, U, U, R, R, abi.F32])}
    F = E.ScalarFunction
    lists = [
        [F("gt", c[0], E.Constant(0))],
        [c[0], c[4]],
        [F("gt", c[0], E.Constant(-50000)), F("lt", F("plus", c[0], c[1]), E.Constant(1000)), c[5]],
        [F("or", F("isnull", c[0]), F("gt", c[1], E.Constant(0))), F("div", c[4], F("minus", c[5], c[5]))],
    ]
    for lst in lists:
        progs = E.compile_list(lst)
        keep = []
        cols = make_cols(chk.columns, keep)
        n = chk.NumRows()
        s = np.zeros(n, np.uint8)
        z = np.zeros(n, np.uint8)
        w = C.c_int64(0)
        st = sim.sim_filter_eval(progs, len(lst), cols, len(chk.columns), n, None, s.ctypes.data_as(C.c_void_p),
                                 z.ctypes.data_as(C.c_void_p), C.byref(w))
        osel, onull, ow = orc.filter_eval(progs, len(lst), chk)
        assert st == abi.OK
        assert (s.astype(bool) == osel).all() and (z.astype(bool) == onull).all() and w.value == ow


def test_first_error_is_first_node_then_first_row(sim, orc):
    # row 0 overflows in node `mul` (later node), row 1 overflows in node `plus` (earlier node):
    # the node-at-a-time evaluator reports `plus` (BIGINT UNSIGNED here) — so must the fused kernel.
    i64max, u64max = (1 << 63) - 1, (1 << 64) - 1
    chk = H.chunk_from_rows([[i64max, 1, 2], [1, u64max, 1]], [I, U, U])
    F = E.ScalarFunction
    e = F("mul", F("mul", E.Column(0, I), F("in", E.Column(0, I), E.Constant(i64max))),
          F("isnull", F("plus", E.Column(1, U), E.Column(2, U))))
    prog = E.compile_expr(e)
    st, _, _, _ = sim_eval(sim, prog, chk)
    with pytest.raises(orc.OracleError) as ei:
        orc.expr_eval(prog, chk)
    assert st == ei.value.status == abi.ERR_OVERFLOW_BIGINT_UNSIGNED


def test_generator_matches_oracle(sim, orc):
    for kind, extra in [(abi.GEN_SEQ, {}), (abi.GEN_AFFINE, dict(a=48271, b=11, m=100003)), (abi.GEN_RAND_MOD, dict(m=1000)),
                        (abi.GEN_RAND_F64, {}), (abi.GEN_HASH_OF_COL, dict(b=0x1234)), (abi.GEN_ZIPF_OCT, dict(a=20, m=1_000_000))]:
        spec = abi.GenSpec()
        spec.kind, spec.table, spec.col, spec.null_pct, spec.seed, spec.start = kind, 3, 1, 7, 42, 1000
        for k, v in extra.items():
            setattr(spec, k, v)
        n = 5000
        src = np.arange(n, dtype=np.uint64) * 977
        dst = np.zeros(n, np.uint64)
        nn = np.zeros(n, np.uint8)
        sim.sim_gen_column(C.byref(spec), n, dst.ctypes.data_as(C.c_void_p), nn.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p))
        od, obm = orc.gen_column(spec, n, src)
        assert (od == dst).all()
        assert (np.unpackbits(obm, bitorder="little")[:n] == nn).all()


def test_affine_is_a_bijection(orc):
    spec = abi.GenSpec()
    spec.kind, spec.seed, spec.a, spec.b, spec.m = abi.GEN_AFFINE, 42, 2654435761, 17, 100000
    d, _ = orc.gen_column(spec, 100000)
    assert len(np.unique(d)) == 100000 and d.max() == 99999


def test_mulhi_and_rank_ranges(sim):
    rng = np.random.default_rng(9)
    for _ in range(200):
        a, b = int(rng.integers(0, 1 << 63)) * 2 + 1, int(rng.integers(1, 1 << 40))
        assert sim.sim_mulhi64(a, b) == (a * b) >> 64
        assert sim.sim_mulhi64(a, b) < b
    for parts in (1, 2, 3, 4, 8, 64):
        ranks = [sim.sim_key_rank(int(k), parts) for k in rng.integers(0, 1 << 62, 2000)]
        assert 0 <= min(ranks) and max(ranks) < parts
        if parts > 1:
            counts = np.bincount(ranks, minlength=parts)
            assert counts.min() > 2000 / parts * 0.6


def test_rowhash_matches_oracle_checksum(sim, orc):
    rows = [[1, None, 2.5], [-7, 3, None]]
    chk = H.chunk_from_rows(rows, [I, I, R])
    s = x = 0
    for r in range(2):
        vals = np.array([chk.columns[c].data.view(np.uint64)[r] for c in range(3)], np.uint64)
        nn = np.array([not chk.columns[c].IsNull(r) for c in range(3)], np.uint8)
        h = sim.sim_rowhash(vals.ctypes.data_as(C.c_void_p), nn.ctypes.data_as(C.c_void_p), 3)
        s = (s + h) & ((1 << 64) - 1)
        x ^= h
    assert (s, x) == orc.rows_checksum(chk)


# ---------------------------------------------------------------- strings: the device interpreter's ETString opcodes on the CPU
def test_string_signatures_device_interpreter_vs_oracle(sim, orc):
    """the same expression set as tests/test_expr_string_gpu.py, evaluated by tsq_eval_row compiled for the host: the per-row string
    semantics of the kernels (references on the stack, byte-wise compares, NULL protocols) against the node-at-a-time oracle"""
    from tinysql_amd.chunk import StrColumn
    from .test_oracle_string_golden import rand_strs
    rng = np.random.default_rng(77)
    n = 4000
    chk = Chunk([StrColumn([None if v is None else v * int(rng.integers(1, 6)) for v in rand_strs(rng, n)]), StrColumn(rand_strs(rng, n)),
                 StrColumn(rand_strs(rng, n, 0.4)), Column(abi.I64, rng.integers(-2, 3, n), rng.random(n) > 0.2)])
    S0, S1, S2, I3 = E.Column(0, abi.BYTES), E.Column(1, abi.BYTES), E.Column(2, abi.BYTES), E.Column(3, abi.I64)
    F, K = E.ScalarFunction, E.Constant
    exprs = [F(op, S0, S1) for op in ("lt", "le", "gt", "ge", "eq", "ne")] + [
        F("lt", S1, K(b"ab\x80")), F("eq", K(""), S2), F("strcmp", S0, S1), F("strcmp", S0, K(None, E.ETString)), F("length", S0), F("isnull", S2),
        F("length", F("ifnull", S2, S0)), F("strcmp", F("if", I3, S0, S1), F("ifnull", S2, K("abc"))),
        F("in", S1, K("a"), K("ab"), K(b"\xff"), K("")), F("in", S1, S2, K(None, E.ETString), S0),
        F("plus", F("length", S0), F("mul", I3, F("strcmp", S1, S2)))]
    sel = np.sort(rng.permutation(n)[: n // 3]).astype(np.int32)
    for e in exprs:
        check_same(sim, orc, e, chk)
        check_same(sim, orc, e, Chunk(chk.columns, sel=sel))


def test_string_program_validation(sim):
    S0 = E.Column(0, abi.BYTES)
    prog = E.compile_expr(E.ScalarFunction("length", S0))
    sim.sim_validate_typed.restype = C.c_int32
    sim.sim_validate_typed.argtypes = [C.POINTER(abi.ExprProg), C.c_int32, C.POINTER(C.c_int32)]
    assert sim.sim_validate_typed(C.byref(prog), 1, (C.c_int32 * 1)(abi.BYTES)) == abi.OK
    assert sim.sim_validate_typed(C.byref(prog), 1, (C.c_int32 * 1)(abi.I64)) == abi.ERR_INVALID       # COL_STR on a BIGINT column
    prog.ops[1].opcode = abi.OP_NEG_INT                                                                 # a number operator on a string value
    assert sim.sim_validate(C.byref(prog), 1) == abi.ERR_INVALID
    prog = E.compile_expr(E.ScalarFunction("length", S0))
    prog.n_ops = 1                                                                                      # a string-valued root ...
    assert sim.sim_validate(C.byref(prog), 1) == abi.ERR_UNSUPPORTED                                    # ... declared as an Int result: the fallback signal
    prog.result_type = abi.BYTES                                                                        # ... declared as such: tsq_expr_eval_str
    assert sim.sim_validate(C.byref(prog), 1) == abi.OK
    prog = E.compile_expr(E.ScalarFunction("ifnull", S0, E.Constant("x")))
    assert prog.result_type == abi.BYTES and sim.sim_validate(C.byref(prog), 1) == abi.OK
    prog.result_type = abi.I64
    assert sim.sim_validate(C.byref(prog), 1) == abi.ERR_UNSUPPORTED
    prog = E.compile_expr(E.ScalarFunction("length", S0))
    prog.result_type = abi.BYTES                                                                        # a numeric root declared TSQ_BYTES is malformed
    assert sim.sim_validate(C.byref(prog), 1) == abi.ERR_INVALID
