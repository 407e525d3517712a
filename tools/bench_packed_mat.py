#!/usr/bin/env python3
"""Materialising join through the packed-key route (csrc/tsq_dajoin.h, K4d): shapes the 64-bit LDS route refuses.
   python tools/bench_packed_mat.py [rows]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    ctx = _lib.Context(0)
    lib = ctx.lib

    def spec(kind, **kw):
        s = abi.GenSpec()
        s.kind, s.seed = kind, 42
        for k, v in kw.items():
            setattr(s, k, v)
        return s

    def col(ptr, rows, bm=None, tp=abi.I64):
        c = abi.Col()
        c.data, c.length, c.elem_size, c.type, c.flags = ptr, rows, 8, tp, abi.COL_DEVICE
        if bm:
            c.null_bitmap = bm
        return c

    bk, pk = ctx.alloc(n * 8), ctx.alloc(n * 8)
    bv = [ctx.alloc(n * 8) for _ in range(5)]
    pv = [ctx.alloc(n * 8) for _ in range(3)]
    bm_b, bm_p, bm_k = ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64)
    ctx.gen_column(spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n), n, bk)
    ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=n), n, pk)
    for i, p in enumerate(bv):
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=2, col=2 + i, m=1 << 40, null_pct=3 if i == 0 else 0), n, p, null_bitmap=bm_b if i == 0 else None)
    for i, p in enumerate(pv):
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=2 + i, m=1 << 40, null_pct=3 if i == 0 else 0), n, p, null_bitmap=bm_p if i == 0 else None)
    tmp = ctx.alloc(n * 8)
    ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=9, m=n, null_pct=3), n, tmp, null_bitmap=bm_k)  # only its bitmap is used: 3 % NULL probe keys
    ctx.free(tmp)
    ctx.sync()
    out = {}
    shapes = {
        "nullable_inner_2x2": (abi.JOIN_INNER, [col(bk, n), col(bv[0], n, bm_b)], [col(pk, n), col(pv[0], n, bm_p)]),
        "nullable_left_outer_2x2": (abi.JOIN_LEFT_OUTER, [col(bk, n), col(bv[0], n, bm_b)], [col(pk, n, bm_k), col(pv[0], n, bm_p)]),
        "wide_inner_6x4": (abi.JOIN_INNER, [col(bk, n)] + [col(p, n) for p in bv], [col(pk, n)] + [col(p, n) for p in pv]),
    }
    only = os.environ.get("SHAPES")
    for name, (jt, bcols, pcols) in shapes.items():
        if only and name not in only.split(","):
            continue
        cfg = abi.JoinCfg()
        cfg.join_type, cfg.build_is_right, cfg.n_keys = jt, 1, 1
        cfg.n_build_cols, cfg.n_probe_cols = len(bcols), len(pcols)
        for i in range(len(bcols)):
            cfg.build_types[i] = abi.I64
        for i in range(len(pcols)):
            cfg.probe_types[i] = abi.I64
        times, rows, st = [], 0, abi.Stats()
        for rep in range(3):  # the output buffers of rep k come from the context's pool once rep k - 1 has been destroyed
            h = C.c_void_p()
            _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
            try:
                _lib.check(lib.tsq_join_build_push(h, (abi.Col * len(bcols))(*bcols), len(bcols), n), h)
                _lib.check(lib.tsq_join_build_finish(h), h)
                ctx.sync()
                t = time.perf_counter()
                _lib.check(lib.tsq_join_probe_push(h, (abi.Col * len(pcols))(*pcols), len(pcols), n, None), h)
                ctx.sync()
                times.append((time.perf_counter() - t) * 1e3)
                _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
                _lib.check(lib.tsq_join_probe_finish(h), h)
                c = C.c_int64(0)
                _lib.check(lib.tsq_join_count(h, C.byref(c)), h)
                rows = c.value
            finally:
                lib.tsq_join_destroy(h)
        ncols = len(bcols) + len(pcols)
        algo = 8.0 * len(pcols) * n + 16.0 * n + 8.0 * ncols * rows  # probe columns read + one slot per probe row + every output cell written
        out[name] = {"ms_reps": times, "ms": min(times[1:]), "joined_rows": rows, "route": st.probe_route, "radix_batches": st.radix_batches,
                     "frac": algo / (min(times[1:]) * 1e-3) / 8e12, "packed_build_ms": st.packed_build_ms, "note": "one probe_push of all rows incl. the packed-key images (first push of a build side)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
