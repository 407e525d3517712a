#!/usr/bin/env python3
"""tsq_sort timing with a STRING ORDER BY item: N device-resident (name char(W), v int64) rows, names = W random bytes each (every
byte position differs between rows: one radix pass per byte) or W-byte decimal-like names (few distinct values per position).
usage: bench_sort_str.py [rows] [width]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
import gpu_helpers as G  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    assert w % 8 == 0
    with _lib.Context(0) as ctx:
        lib = ctx.lib
        data = ctx.alloc(n * w + 64)
        offs = ctx.alloc((n + 1) * 8 + 64)
        v = G.DevCol(ctx, abi.I64, n)
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, a=w, b=0, m=(1 << 31) - 1), n + 1, offs)  # offsets = w * i
        ctx.gen_column(G.gen_spec(abi.GEN_SEQ), n, v.data)
        od, oo, ob = ctx.alloc(n * w + 64), ctx.alloc((n + 1) * 8 + 64), ctx.alloc(n // 8 + 64)
        ov = G.DevCol(ctx, abi.I64, n, with_nulls=True)
        for label, mod in (("%d random bytes" % w, 1 << 62), ("%d bytes, 4 distinct values per 8-byte word" % w, 4)):
            ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=6, col=0, m=mod), n * w // 8, data)
            cfg = abi.SortCfg()
            cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count = 2, 1, 0, -1
            cfg.col_types[0], cfg.col_types[1] = abi.BYTES, abi.I64
            name = abi.Col()
            name.data, name.offsets, name.length, name.elem_size, name.type, name.flags = data, offs, n, -1, abi.BYTES, abi.COL_DEVICE
            best, best_pull, passes = 1e30, 1e30, 0
            for rep in range(2):
                h = C.c_void_p()
                _lib.check(lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
                try:
                    _lib.check(lib.tsq_sort_push(h, (abi.Col * 2)(name, v.col()), 2, n), h)
                    ctx.sync()
                    t = time.perf_counter()
                    _lib.check(lib.tsq_sort_finish(h), h)
                    ctx.sync()
                    best = min(best, time.perf_counter() - t)
                    out = abi.Col()
                    out.data, out.offsets, out.null_bitmap, out.length, out.elem_size, out.type, out.flags = od, oo, ob, n, -1, abi.BYTES, abi.COL_DEVICE
                    t = time.perf_counter()
                    m, eos = C.c_int64(0), C.c_int32(0)
                    _lib.check(lib.tsq_sort_pull(h, (abi.Col * 2)(out, ov.col()), 2, n, C.byref(m), C.byref(eos)), h)
                    ctx.sync()
                    best_pull = min(best_pull, time.perf_counter() - t)
                    rows, p, sk, ms = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
                    _lib.check(lib.tsq_sort_stats(h, C.byref(rows), C.byref(p), C.byref(sk), C.byref(ms)), h)
                    passes = p.value
                finally:
                    lib.tsq_sort_destroy(h)
            head = np.zeros((1 << 16) * w, np.uint8)
            ctx.d2h(head, od)
            s = [head[i * w:(i + 1) * w].tobytes() for i in range(1 << 16)]
            assert all(s[i] <= s[i + 1] for i in range(len(s) - 1))
            print(json.dumps({"keys": label, "rows": n, "digit_passes": passes, "sort_finish_ms": best * 1e3, "gather_pull_ms": best_pull * 1e3, "rows_per_s": n / best}))


if __name__ == "__main__":
    main()
