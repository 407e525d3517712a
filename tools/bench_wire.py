#!/usr/bin/env python3
"""tsq_chunk_encode / tsq_chunk_decode timing: a device chunk of (bigint with NULLs, double, bigint, varchar) <-> its wire buffer in HBM
(chunk.Codec, util/chunk/codec.go:42-143).  usage: bench_wire.py [rows]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402


def col(data, bitmap, offsets, n, tp):
    c = abi.Col()
    c.data, c.null_bitmap, c.offsets, c.length = data, bitmap, offsets, n
    c.elem_size, c.type, c.flags = (-1 if tp == abi.BYTES else 8), tp, abi.COL_DEVICE
    return c


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
    with _lib.Context(0) as ctx:
        lib = ctx.lib
        types = [abi.I64, abi.F64, abi.I64, abi.BYTES]
        d = [ctx.alloc(8 * n + 64) for _ in range(3)]
        bm = ctx.alloc(n // 8 + 64)
        s = abi.GenSpec()
        s.kind, s.seed, s.table, s.col, s.m, s.null_pct = abi.GEN_RAND_MOD, 7, 3, 0, 1 << 62, 10
        ctx.gen_column(s, n, d[0], bm)
        s.null_pct, s.col = 0, 1
        ctx.gen_column(s, n, d[1])
        s.kind, s.start = abi.GEN_SEQ, 0
        ctx.gen_column(s, n, d[2])
        # the var-len column: 8-byte cells (its data = column 1's bytes, offsets = 8 i)
        offs = ctx.alloc(8 * (n + 1) + 64)
        s.kind, s.start = abi.GEN_AFFINE, 0
        s.a, s.b, s.m = 8, 0, (1 << 31) - 1  # 8 i stays below m for n <= 2.6e8
        ctx.gen_column(s, n + 1, offs)
        cols = (abi.Col * 4)(col(d[0], bm, None, n, abi.I64), col(d[1], None, None, n, abi.F64), col(d[2], None, None, n, abi.I64),
                             col(d[1], None, offs, n, abi.BYTES))
        need = C.c_int64(0)
        _lib.check(lib.tsq_chunk_encode(ctx.h, cols, 4, n, None, 0, abi.COL_DEVICE, C.byref(need)), ctx.h)
        wire = ctx.alloc(need.value + 64)
        od = [ctx.alloc(8 * n + 64) for _ in range(4)]
        obm = [ctx.alloc(n // 8 + 64) for _ in range(4)]
        ooffs = ctx.alloc(8 * (n + 1) + 64)
        tp = (C.c_int32 * 4)(*types)
        nrows, used = C.c_int64(0), C.c_int64(0)

        def encode():
            return lib.tsq_chunk_encode(ctx.h, cols, 4, n, C.c_void_p(wire), need.value, abi.COL_DEVICE, C.byref(need))

        def decode():
            out = (abi.Col * 4)(*[col(od[i], obm[i], ooffs if types[i] == abi.BYTES else None, 0, types[i]) for i in range(4)])
            return lib.tsq_chunk_decode(ctx.h, C.c_void_p(wire), need.value, abi.COL_DEVICE, tp, 4, 0, n, out, C.byref(nrows), C.byref(used))

        res = {"rows": n, "wire_bytes": need.value}
        for name, call in (("encode", encode), ("decode", decode)):
            _lib.check(call(), ctx.h)
            ctx.sync()
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                _lib.check(call(), ctx.h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
            res[name + "_ms"] = best * 1e3
            res[name + "_GBs"] = 2.0 * need.value / best / 1e9   # every wire byte is read once and written once
            res[name + "_frac"] = 2.0 * need.value / best / 8e12
        a, b = np.zeros(1 << 20, np.int64), np.zeros(1 << 20, np.int64)
        ctx.d2h(a, d[1])
        ctx.d2h(b, od[3])
        res["verified"] = bool(nrows.value == n and used.value == need.value and (a == b).all())
        print(json.dumps(res))


if __name__ == "__main__":
    main()
