#!/usr/bin/env python3
"""Materialising inner join timing: N probe rows x N build rows of (k int64, v int64), every probe row joins once, the joined rows
(4 columns) stay in HBM.  usage: bench_emit.py [rows]"""
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


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    with _lib.Context(0) as ctx:
        lib = ctx.lib
        bk, bv, pk, pv = (G.DevCol(ctx, abi.I64, n) for _ in range(4))
        outs = [G.DevCol(ctx, abi.I64, n, with_nulls=True) for _ in range(4)]
        try:
            ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n), n, bk.data)
            ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=2, col=1, m=1 << 30), n, bv.data)
            ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=n), n, pk.data)
            ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 30), n, pv.data)
            cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
            best, best_pull = 1e30, 1e30
            for rep in range(3):
                h = C.c_void_p()
                _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
                try:
                    _lib.check(lib.tsq_join_build_push(h, G.dev_cols([bk, bv]), 2, n), h)
                    _lib.check(lib.tsq_join_build_finish(h), h)
                    ctx.sync()
                    t = time.perf_counter()
                    _lib.check(lib.tsq_join_probe_push(h, G.dev_cols([pk, pv]), 2, n, None), h)
                    _lib.check(lib.tsq_join_probe_finish(h), h)
                    ctx.sync()
                    best = min(best, time.perf_counter() - t)
                    t = time.perf_counter()
                    total = 0
                    while True:
                        m, eos = C.c_int64(0), C.c_int32(0)
                        _lib.check(lib.tsq_join_pull(h, G.dev_cols(outs), 4, n, C.byref(m), C.byref(eos)), h)
                        if m.value == 0:
                            break
                        total += m.value
                    ctx.sync()
                    best_pull = min(best_pull, time.perf_counter() - t)
                    assert total == n
                finally:
                    lib.tsq_join_destroy(h)
            algo = n * 32.0 + n * 24.0
            print(json.dumps({"workload": "materialising inner join %d x %d (k, v), hit ratio 1.0, result in HBM" % (n, n), "probe_ms": best * 1e3, "pull_copy_ms": best_pull * 1e3,
                              "joined_rows_per_s": n / best, "algorithmic_GBs": algo / best / 1e9, "frac_of_8TBs": algo / best / 8e12}))
        finally:
            for d in [bk, bv, pk, pv] + outs:
                d.free()


if __name__ == "__main__":
    main()
