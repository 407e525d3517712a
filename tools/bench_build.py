#!/usr/bin/env python3
"""Build-side timing: tsq_join_build_finish on N device-resident (k int64, v int64) rows, partitioned (LDS slice images)
vs row-at-a-time CAS.  usage: bench_build.py [N ...]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
import gpu_helpers as G  # noqa: E402
import helpers as H  # noqa: E402


def one(ctx, n, radix):
    lib = ctx.lib
    bk, bv = G.DevCol(ctx, abi.I64, n), G.DevCol(ctx, abi.I64, n)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n), n, bk.data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=2, col=1, m=1 << 30), n, bv.data)
        cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
        best_wall, best_ev, st = 1e30, 1e30, None
        for rep in range(3):
            h = C.c_void_p()
            _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
            try:
                _lib.check(lib.tsq_join_set_radix(h, radix), h)
                _lib.check(lib.tsq_join_build_push(h, G.dev_cols([bk, bv]), 2, n), h)
                ctx.sync()
                t = time.perf_counter()
                _lib.check(lib.tsq_join_build_finish(h), h)
                ctx.sync()
                best_wall = min(best_wall, time.perf_counter() - t)
                st = abi.Stats()
                _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
                best_ev = min(best_ev, st.build_kernel_ms)
            finally:
                lib.tsq_join_destroy(h)
        return {"rows": n, "partitioned": int(st.build_partitioned), "build_finish_ms": best_wall * 1e3, "build_kernels_ms": best_ev,
                "rows_per_s": n / (best_ev * 1e-3), "table_bytes": int(st.table_bytes), "handed_back_rows": int(st.build_handed_back_rows),
                "roofline_frac_32B_per_row": 32.0 * n / (best_ev * 1e-3) / 8e12}
    finally:
        bk.free()
        bv.free()


if __name__ == "__main__":
    sizes = [int(float(x)) for x in sys.argv[1:]] or [10_000_000, 100_000_000]
    with _lib.Context(0) as ctx:
        for n in sizes:
            for radix in (abi.RADIX_AUTO, abi.RADIX_OFF):
                print(json.dumps(one(ctx, n, radix)))
