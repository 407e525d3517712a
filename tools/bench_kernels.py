#!/usr/bin/env python3
"""Per-kernel roofline measurements for the SURVEY §8(d) rows that bench.py's single line does not carry:
build, materialising probe, expression / filter evaluation, multi-GPU split.  Device-resident inputs, HIP-event timing.
One JSON object per line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
from tinysql_amd import expression as E  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import dev_col, spec  # noqa: E402


def timed(ctx, fn, reps=3):
    best = 1e30
    for i in range(reps):
        if os.environ.get("BK_TRACE"):
            print("  rep", i, "start", file=sys.stderr, flush=True)
        ctx.timer_start()
        fn()
        best = min(best, ctx.timer_stop_ms())
        if os.environ.get("BK_TRACE"):
            print("  rep", i, "done", file=sys.stderr, flush=True)
    return best


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 100_000_000
    with _lib.Context(0) as ctx:
        lib = ctx.lib
        a, b, c, out = (ctx.alloc(n * 8) for _ in range(4))
        bm = ctx.alloc(n // 8 + 64)
        sel = ctx.alloc(n + 64)
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=5, col=0, m=1 << 20), n, a)
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=5, col=1, m=1 << 20), n, b)
        ctx.gen_column(spec(abi.GEN_RAND_F64, table=5, col=2), n, c)
        cols = (abi.Col * 3)(dev_col(a, n), dev_col(b, n), dev_col(c, n, abi.F64))
        # ---- expression: (a + b) * 3 - a   (int64, overflow checked at every node)
        e1 = E.ScalarFunction("minus", E.ScalarFunction("mul", E.ScalarFunction("plus", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.Constant(3)), E.Column(0, abi.I64))
        ce = E.CompiledExpr(ctx, [e1], jit=abi.JIT_FORCE if '--no-jit' not in sys.argv else abi.JIT_OFF)
        oc = dev_col(out, n)
        oc.null_bitmap = bm
        w = C.c_int64(0)
        ms = timed(ctx, lambda: _lib.check(lib.tsq_expr_eval(ce.h, cols, 3, n, None, C.byref(oc), C.byref(w)), ce.h))
        byt = n * 8 * 2 + n * 8
        print(json.dumps({"kernel": "k_expr_eval (a+b)*3-a int64", "rows": n, "ms": ms, "algorithmic_GBs": byt / ms / 1e6, "frac_of_8TBs": byt / ms / 1e6 / 8000}), flush=True)
        print(json.dumps({"jit_launches(expr)": ce.jit_launches()}), flush=True)
        ce.close()
        # ---- filter: a < b AND c > 0.5
        f = [E.ScalarFunction("lt", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.ScalarFunction("gt", E.Column(2, abi.F64), E.Constant(0.5))]
        cf = E.CompiledExpr(ctx, f, jit=abi.JIT_FORCE if '--no-jit' not in sys.argv else abi.JIT_OFF)
        ms = timed(ctx, lambda: _lib.check(lib.tsq_filter_eval(cf.h, cols, 3, n, None, sel, None, C.byref(w)), cf.h))
        byt = n * 8 * 3 + n
        print(json.dumps({"kernel": "k_filter_eval a<b AND c>0.5", "rows": n, "ms": ms, "algorithmic_GBs": byt / ms / 1e6, "frac_of_8TBs": byt / ms / 1e6 / 8000}), flush=True)
        cf.close()
        # ---- build + materialising probe: n/4 probe rows x n/4 build rows, (k, v) each side
        nb = npr = n // 4
        cfg = abi.JoinCfg()
        cfg.join_type, cfg.build_is_right, cfg.n_keys, cfg.n_build_cols, cfg.n_probe_cols = abi.JOIN_INNER, 1, 1, 2, 2
        for i in range(2):
            cfg.build_types[i] = cfg.probe_types[i] = abi.I64
        ctx.gen_column(spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=nb), nb, a)
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=nb), npr, b)
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        _lib.check(lib.tsq_join_build_push(h, (abi.Col * 2)(dev_col(a, nb), dev_col(c, nb)), 2, nb), h)
        _lib.check(lib.tsq_join_build_finish(h), h)
        st = abi.Stats()
        ctx.timer_start()
        _lib.check(lib.tsq_join_probe_push(h, (abi.Col * 2)(dev_col(b, npr), dev_col(out, npr)), 2, npr, None), h)
        _lib.check(lib.tsq_join_probe_finish(h), h)
        ms = ctx.timer_stop_ms()
        _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
        byt_b = 32.0 * nb
        print(json.dumps({"kernel": "k_build_insert", "rows": nb, "ms": st.build_kernel_ms, "algorithmic_GBs": byt_b / st.build_kernel_ms / 1e6,
                          "frac_of_8TBs": byt_b / st.build_kernel_ms / 1e6 / 8000}), flush=True)
        byt_p = npr * 32.0 + st.out_rows * 24.0
        print(json.dumps({"kernel": "materialising probe (k_probe_count + k_probe_emit, incl. host bookkeeping)", "probe_rows": npr, "out_rows": st.out_rows, "ms": ms,
                          "algorithmic_GBs": byt_p / ms / 1e6, "frac_of_8TBs": byt_p / ms / 1e6 / 8000}), flush=True)
        lib.tsq_join_destroy(h)
        # ---- split into 8 parts (key + one payload)
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=1 << 40), n, a)
        counts = (C.c_int64 * 8)()
        ms = timed(ctx, lambda: _lib.check(lib.tsq_radix_split(ctx.h, (abi.Col * 2)(dev_col(a, n), dev_col(b, n)), 2, 0, 0, n, 8, (abi.Col * 2)(dev_col(c, n), dev_col(out, n)), counts), ctx.h))
        byt = n * 8.0 + n * 32.0  # histogram pass reads the key, partition pass moves key + payload both ways
        print(json.dumps({"kernel": "tsq_radix_split 8 parts (k_rank_hist + k_radix_partition<V=1>)", "rows": n, "ms": ms, "algorithmic_GBs": byt / ms / 1e6, "frac_of_8TBs": byt / ms / 1e6 / 8000}), flush=True)
        for p in (a, b, c, out, bm, sel):
            ctx.free(p)


if __name__ == "__main__":
    main()
