#!/usr/bin/env python3
"""tsq_sort timing: ORDER BY k over N device-resident (k int64, v int64) rows — full-range keys (8 digit passes) and
day-number keys (2 passes) — and the CPU baseline (oracle restatement of SortExec, std::stable_sort, one thread).
usage: bench_sort.py [rows]"""
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
from tinysql_amd.chunk import Chunk, Column  # noqa: E402
import gpu_helpers as G  # noqa: E402


def run(ctx, n, mod, label, limit=-1):
    k, v = G.DevCol(ctx, abi.I64, n), G.DevCol(ctx, abi.I64, n)
    ok_, ov = G.DevCol(ctx, abi.I64, n, with_nulls=True), G.DevCol(ctx, abi.I64, n, with_nulls=True)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=5, col=0, m=mod), n, k.data)
        ctx.gen_column(G.gen_spec(abi.GEN_SEQ), n, v.data)
        cfg = abi.SortCfg()
        cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count = 2, 1, 0, limit
        cfg.col_types[0] = cfg.col_types[1] = abi.I64
        best, best_k, best_pull, passes = 1e30, 1e30, 1e30, 0
        for rep in range(3):
            h = C.c_void_p()
            _lib.check(ctx.lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
            try:
                _lib.check(ctx.lib.tsq_sort_push(h, G.dev_cols([k, v]), 2, n), h)
                ctx.sync()
                t = time.perf_counter()
                _lib.check(ctx.lib.tsq_sort_finish(h), h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
                t = time.perf_counter()
                m, eos = C.c_int64(0), C.c_int32(0)
                _lib.check(ctx.lib.tsq_sort_pull(h, G.dev_cols([ok_, ov]), 2, n, C.byref(m), C.byref(eos)), h)
                ctx.sync()
                best_pull = min(best_pull, time.perf_counter() - t)
                rows, p, sk, ms = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
                _lib.check(ctx.lib.tsq_sort_stats(h, C.byref(rows), C.byref(p), C.byref(sk), C.byref(ms)), h)
                best_k, passes = min(best_k, ms.value), p.value
            finally:
                ctx.lib.tsq_sort_destroy(h)
        keys = ok_.to_host().data[: (n if limit < 0 else limit)]
        assert (np.diff(keys[: 1 << 22]) >= 0).all()
        algo = 24.0 * n * max(passes, 1)
        return {"keys": label + ("" if limit < 0 else ", TopN LIMIT %d (radix select + sort of the candidates)" % limit), "rows": n, "digit_passes": passes, "sort_finish_ms": best * 1e3, "sort_kernels_ms": best_k, "gather_pull_ms": best_pull * 1e3,
                "rows_per_s": n / best, "algorithmic_GBs": algo / (best_k * 1e-3) / 1e9, "frac_of_8TBs": algo / (best_k * 1e-3) / 8e12}
    finally:
        for d in (k, v, ok_, ov):
            d.free()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    rng = np.random.default_rng(1)
    m = 5_000_000
    chk = Chunk([Column(abi.I64, rng.integers(0, 1 << 40, m)), Column(abi.I64, np.arange(m))])
    from oracle import binding as orc  # cpu_baseline leg only: the oracle's SortExec restatement (test infrastructure)
    t = time.perf_counter()
    orc.sort_perm(chk, [0], [False])
    cpu_s = time.perf_counter() - t
    with _lib.Context(0) as ctx:
        for mod, label in ((1 << 62, "uniform 62-bit"), (1 << 40, "uniform 40-bit"), (2500, "day numbers 0..2499")):
            r = run(ctx, n, mod, label)
            r["cpu_baseline"] = {"kind": "port", "cores": 1, "rows_per_s": m / cpu_s, "sample": "oracle SortExec restatement (stable_sort), %d rows, 40-bit keys" % m}
            print(json.dumps(r))
        r = run(ctx, n, 1 << 62, "uniform 62-bit", limit=100)
        print(json.dumps(r))


if __name__ == "__main__":
    main()
