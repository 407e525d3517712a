"""Shared helpers for the test-suite: golden-case lowering, random tables, multiset comparison."""
import json
import math
import os

import numpy as np

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column, StrColumn, np_dtype

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TYPES = {"i64": abi.I64, "u64": abi.U64, "f32": abi.F32, "f64": abi.F64, "str": abi.BYTES}
JOIN_TYPES = {"inner": abi.JOIN_INNER, "left": abi.JOIN_LEFT_OUTER, "right": abi.JOIN_RIGHT_OUTER}
AGG_FUNCS = {"count": abi.AGG_COUNT, "sum": abi.AGG_SUM, "avg": abi.AGG_AVG, "max": abi.AGG_MAX, "min": abi.AGG_MIN,
             "firstrow": abi.AGG_FIRSTROW}


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def chunk_from_rows(rows, types):
    """rows: list of lists with None for NULL; types: list of abi type codes."""
    cols = []
    for c, tp in enumerate(types):
        vals = [r[c] for r in rows]
        if tp == abi.BYTES:
            cols.append(StrColumn(vals))
            continue
        nn = np.array([v is not None for v in vals], dtype=bool)
        if tp == abi.U64:
            data = np.array([0 if v is None else int(v) for v in vals], dtype=np.uint64)
        else:
            data = np.array([0 if v is None else v for v in vals], dtype=np_dtype(tp))
        cols.append(Column(tp, data.reshape(-1), nn if len(vals) else None))
    return Chunk(cols)


def canon(v):
    """canonical hashable form of a cell (floats by bit pattern so -0.0/NaN compare exactly)."""
    if v is None:
        return None
    if isinstance(v, float):
        return ("f", np.float64(v).view(np.uint64).item())
    if isinstance(v, (bytes, bytearray)):
        return ("b", bytes(v))
    return int(v)


def multiset(chunk_or_rows):
    rows = chunk_or_rows.rows() if hasattr(chunk_or_rows, "rows") else chunk_or_rows
    return sorted((tuple(canon(v) for v in r) for r in rows), key=lambda t: tuple((x is None, str(x)) for x in t))


def rows_equal_unordered(a, b):
    """multiset equality of rows (join / aggregate output order is unspecified).  Two big fixed-width chunks are compared by
    (row count, sum and xor of the oracle's 64-bit row hashes) — SURVEY.md 8(d)'s multiset fingerprint — computed by the oracle's C++
    (orc_rows_checksum) on BOTH sides: the GPU suite spent minutes building Python tuples of million-row results (698 s of
    a 1200 s limit).  A fingerprint mismatch falls through to the explicit comparison, so a failure still shows rows."""
    if hasattr(a, "columns") and hasattr(b, "columns") and a.sel is None and b.sel is None and a.NumRows() == b.NumRows() > 20000:
        ta, tb = a.types(), b.types()
        if ta == tb and abi.BYTES not in ta:
            from oracle import binding as orc
            if orc.rows_checksum(a) == orc.rows_checksum(b):
                return True
    return multiset(a) == multiset(b)


def expr_from_json(j, col_types):
    """["col", i] | ["const", v] | ["uconst", v] | ["constnull", "int"|"real"] | [fn, args...]"""
    op = j[0]
    if op == "col":
        return E.Column(j[1], col_types[j[1]])
    if op == "const":
        return E.Constant(j[1])
    if op == "uconst":
        return E.Constant(j[1], E.ETInt, unsigned=True)
    if op == "constnull":
        return E.Constant(None, E.ETReal if j[1] == "real" else E.ETInt)
    return E.ScalarFunction(op, *[expr_from_json(a, col_types) for a in j[1:]])


def join_cfg(left_types, right_types, left_keys, right_keys, join_type, inner_child, other_conds=(), outer_filter=(),
             keep=None, max_chunk_size=1024, probe_batch_rows=0):
    """builds abi.JoinCfg the way executorBuilder.buildHashJoin does (executor/builder.go:431-484)."""
    cfg = abi.JoinCfg()
    build_is_right = inner_child == 1
    btypes, ptypes = (right_types, left_types) if build_is_right else (left_types, right_types)
    bkeys, pkeys = (right_keys, left_keys) if build_is_right else (left_keys, right_keys)
    cfg.join_type = join_type
    cfg.build_is_right = 1 if build_is_right else 0
    cfg.n_keys = len(bkeys)
    for i in range(len(bkeys)):
        cfg.build_key_idx[i] = bkeys[i]
        cfg.probe_key_idx[i] = pkeys[i]
    cfg.n_build_cols, cfg.n_probe_cols = len(btypes), len(ptypes)
    for i, t in enumerate(btypes):
        cfg.build_types[i] = t
    for i, t in enumerate(ptypes):
        cfg.probe_types[i] = t
    cfg.max_chunk_size = max_chunk_size
    cfg.concurrency = 5
    cfg.probe_batch_rows = probe_batch_rows
    if other_conds:
        arr = E.compile_list(list(other_conds))
        if keep is not None:
            keep.append(arr)
        cfg.other_conds = arr
        cfg.n_other_conds = len(other_conds)
    if outer_filter:
        arr = E.compile_list(list(outer_filter))
        if keep is not None:
            keep.append(arr)
        cfg.outer_filters = arr
        cfg.n_outer_filters = len(outer_filter)
    return cfg


def lower_join_case(case, keep):
    """golden join case -> (cfg, left chunk, right chunk, build chunk, probe chunk, exprs)."""
    lt = [abi.I64] * (len(case["left"][0]) if case["left"] else 1)
    rt = [abi.I64] * (len(case["right"][0]) if case["right"] else 1)
    left, right = chunk_from_rows(case["left"], lt), chunk_from_rows(case["right"], rt)
    inner = case["inner_child"]
    probe_types = lt if inner == 1 else rt
    conds = [expr_from_json(e, lt + rt) for e in case.get("other_conds", [])]
    filt = [expr_from_json(e, probe_types) for e in case.get("outer_filter", [])]
    cfg = join_cfg(lt, rt, case["left_keys"], case["right_keys"], JOIN_TYPES[case["type"]], inner, conds, filt, keep)
    build, probe = (right, left) if inner == 1 else (left, right)
    return cfg, left, right, build, probe, conds, filt


def agg_cfg(in_types, group_by, aggs, est_groups=0):
    """aggs: list of (func, arg_col, arg_type, mode, arg_col2)."""
    cfg = abi.AggCfg()
    cfg.n_group_keys = len(group_by)
    for i, c in enumerate(group_by):
        cfg.group_key_col[i] = c
        cfg.group_key_type[i] = in_types[c]
    cfg.n_aggs = len(aggs)
    for i, a in enumerate(aggs):
        func, arg_col, arg_type = a[0], a[1], a[2]
        mode = a[3] if len(a) > 3 else abi.MODE_COMPLETE
        arg_col2 = a[4] if len(a) > 4 else -1
        cfg.aggs[i].func, cfg.aggs[i].mode, cfg.aggs[i].arg_col = func, mode, arg_col
        cfg.aggs[i].arg_col2, cfg.aggs[i].arg_type = arg_col2, arg_type
    cfg.n_input_cols = len(in_types)
    for i, t in enumerate(in_types):
        cfg.input_types[i] = t
    cfg.est_groups = est_groups
    cfg.max_chunk_size = 1024
    return cfg


def random_column(rng, tp, n, null_frac=0.2, lo=None, hi=None):
    """expression/bench_test.go:56-81 defaultGener: 20% NULL, +-Int63 ints, +-1e6*U(0,1) reals."""
    nn = rng.random(n) >= null_frac if null_frac > 0 else None
    if tp in (abi.F32, abi.F64):
        data = (rng.random(n) * 1e6 * np.where(rng.random(n) < 0.5, -1.0, 1.0)).astype(np_dtype(tp))
    elif lo is not None:
        data = rng.integers(lo, hi, n, dtype=np.int64).astype(np_dtype(tp))
    elif tp == abi.U64:
        data = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    else:
        data = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64)
    return Column(tp, data, nn)


def approx_equal(a, b, tol):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, str):  # golden strings vs the bytes of a var-len cell
        a = a.encode()
    if isinstance(b, str):
        b = b.encode()
    if isinstance(a, float) or isinstance(b, float):
        if math.isnan(a) and math.isnan(b):
            return True
        return abs(a - b) <= tol
    return a == b


_M64 = (1 << 64) - 1


def mix64(k):
    """tsq_mix64 (murmur3 finaliser) — the join table stores mix64(key word) (tinysql_amd/csrc/tsq_jointable.h)."""
    k &= _M64
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & _M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & _M64
    k ^= k >> 33
    return k


def unmix64(w):
    """inverse of mix64 (it is a bijection on 64-bit words): the int64 key whose table word is w."""
    inv1, inv2 = pow(0xFF51AFD7ED558CCD, -1, 1 << 64), pow(0xC4CEB9FE1A85EC53, -1, 1 << 64)
    w &= _M64
    w ^= w >> 33
    w = (w * inv2) & _M64
    w ^= w >> 33
    w = (w * inv1) & _M64
    w ^= w >> 33
    assert mix64(w) == (mix64(w) & _M64)
    return int(np.uint64(w).astype(np.int64))
