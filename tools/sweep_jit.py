"""A/B sweep of jit_expr's forms (TSQ_KNOB_JIT_VARIANT, include/tsq.h) on (a + b) * 3 - a over 1e8 BIGINT rows: one JSON line per variant
(kernel + its launch through tsq_expr_eval, best of 5 on the context's HIP-event timer; every row compared with numpy).
  python tools/sweep_jit.py [variant ...]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tinysql_amd import _abi as abi, _lib, expression as E  # noqa: E402
from tools.bench_sides import _dev_col, _spec  # noqa: E402


def main():
    do_filter = "--filter" in sys.argv
    do_arena = "--arena" in sys.argv  # tsq_ctx_reserve(64 GB) first, as bench.py does: the columns then come out of the context's slab
    variants = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 15, 16, 32, 48, 64, 4 | 32, 12 | 32, 8 | 32]
    n = 100_000_000
    ctx = _lib.Context(0)
    if do_arena:
        ctx.reserve(64 << 30)
    lib = ctx.lib
    a, b, out = (ctx.alloc(n * 8) for _ in range(3))
    bm = ctx.alloc(n // 8 + 64)
    ctx.gen_column(_spec(abi, abi.GEN_RAND_MOD, table=5, col=0, m=1 << 20), n, a)
    ctx.gen_column(_spec(abi, abi.GEN_RAND_MOD, table=5, col=1, m=1 << 20), n, b)
    ctx.sync()
    ha, hb = np.empty(n, np.int64), np.empty(n, np.int64)
    ctx.d2h(ha, a)
    ctx.d2h(hb, b)
    want = (ha + hb) * 3 - ha
    cols = (abi.Col * 2)(_dev_col(abi, a, n), _dev_col(abi, b, n))
    if do_filter:  # a < b AND c > 0.5 (jit_filter; bits 128 / 256 of the variant: non-temporal loads / store)
        c, sel = ctx.alloc(n * 8), ctx.alloc(n + 64)
        ctx.gen_column(_spec(abi, abi.GEN_RAND_F64, table=5, col=2), n, c)
        ctx.sync()
        hc = np.empty(n, np.float64)
        ctx.d2h(hc, c)
        wantf = (ha < hb) & (hc > 0.5)
        cols3 = (abi.Col * 3)(_dev_col(abi, a, n), _dev_col(abi, b, n), _dev_col(abi, c, n, abi.F64))
        f = [E.ScalarFunction("lt", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.ScalarFunction("gt", E.Column(2, abi.F64), E.Constant(0.5))]
        wf = C.c_int64(0)
        for v in variants:
            ctx.set_knob(abi.KNOB_JIT_VARIANT, v)
            cf = E.CompiledExpr(ctx, f, jit=abi.JIT_FORCE)
            try:
                fn = lambda: _lib.check(lib.tsq_filter_eval(cf.h, cols3, 3, n, None, sel, None, C.byref(wf)), cf.h)  # noqa: E731
                fn()
                ts = []
                for _ in range(7):
                    ctx.timer_start()
                    fn()
                    ts.append(ctx.timer_stop_ms())
                hs = np.empty(n, np.uint8)
                ctx.d2h(hs, sel)
                print(json.dumps({"filter_variant": v, "ms_min": round(min(ts), 4), "ms_med": round(sorted(ts)[3], 4), "frac": round(25.0 * n / min(ts) / 1e6 / 8000.0, 4),
                                  "ok": bool((hs.astype(bool) == wantf).all())}), flush=True)
                ctx.memset(sel, 0, n)
            finally:
                cf.close()
        ctx.set_knob(abi.KNOB_JIT_VARIANT)
        return
    e1 = E.ScalarFunction("minus", E.ScalarFunction("mul", E.ScalarFunction("plus", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.Constant(3)), E.Column(0, abi.I64))
    w = C.c_int64(0)
    for v in variants:
        ctx.set_knob(abi.KNOB_JIT_VARIANT, v)
        ce = E.CompiledExpr(ctx, [e1], jit=abi.JIT_FORCE)
        try:
            oc = _dev_col(abi, out, n)
            oc.null_bitmap = bm
            fn = lambda: _lib.check(lib.tsq_expr_eval(ce.h, cols, 2, n, None, C.byref(oc), C.byref(w)), ce.h)  # noqa: E731
            fn()
            ts = []
            for _ in range(7):
                ctx.timer_start()
                fn()
                ts.append(ctx.timer_stop_ms())
            ho = np.empty(n, np.int64)
            ctx.d2h(ho, out)
            hbm = np.empty(n // 8, np.uint8)
            ctx.d2h(hbm, bm)
            ok = bool((ho == want).all() and (hbm == 0xff).all())
            print(json.dumps({"variant": v, "ms_min": round(min(ts), 4), "ms_med": round(sorted(ts)[3], 4), "frac": round(24.0 * n / min(ts) / 1e6 / 8000.0, 4), "ok": ok,
                              "jit_launches": ce.jit_launches()}), flush=True)
            ctx.memset(out, 0, n * 8)
            ctx.memset(bm, 0, n // 8)
        finally:
            ce.close()
    ctx.set_knob(abi.KNOB_JIT_VARIANT)


if __name__ == "__main__":
    main()
