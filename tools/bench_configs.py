#!/usr/bin/env python3
"""Side measurements for the BASELINE.json configs that are NOT the bench.py line (they are parity-test cases):
  C2  1e8 x 1e7 int64-key inner hash join, single MI355X, build side resident in HBM
  C3  SELECT k, SUM(v), COUNT(*) GROUP BY k on N rows / 1e6 int64 groups, single MI355X HashAggExec
Prints one JSON object per config.  usage: bench_configs.py [--agg-rows 1e9] [--skip-join] [--skip-agg]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402


def spec(kind, **kw):
    s = abi.GenSpec()
    s.kind, s.seed = kind, 42
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def dev_col(ptr, n, tp=abi.I64):
    c = abi.Col()
    c.data, c.length, c.elem_size, c.type, c.flags = ptr, n, 8, tp, abi.COL_DEVICE
    return c


def join_c2(ctx, nb, npr, steps, radix):
    lib = ctx.lib
    bk, pk = ctx.alloc(nb * 8), ctx.alloc(npr * 8)
    ctx.gen_column(spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=nb), nb, bk)
    ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=nb), npr, pk)
    cfg = abi.JoinCfg()
    cfg.join_type, cfg.build_is_right, cfg.n_keys, cfg.n_build_cols, cfg.n_probe_cols = abi.JOIN_INNER, 1, 1, 1, 1
    cfg.build_types[0] = cfg.probe_types[0] = abi.I64
    h = C.c_void_p()
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    _lib.check(lib.tsq_join_set_radix(h, radix), h)
    _lib.check(lib.tsq_join_build_push(h, (abi.Col * 1)(dev_col(bk, nb)), 1, nb), h)
    _lib.check(lib.tsq_join_build_finish(h), h)
    _lib.check(lib.tsq_join_set_count_only(h, 1), h)
    pc = (abi.Col * 1)(dev_col(pk, npr))
    for _ in range(2):
        _lib.check(lib.tsq_join_probe_push(h, pc, 1, npr, None), h)
    ctx.sync()
    ctx.timer_start()
    for _ in range(steps):
        _lib.check(lib.tsq_join_probe_push(h, pc, 1, npr, None), h)
    ms = ctx.timer_stop_ms() / steps
    cnt = C.c_int64(0)
    _lib.check(lib.tsq_join_count(h, C.byref(cnt)), h)
    st = abi.Stats()
    _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
    lib.tsq_join_destroy(h)
    ctx.free(bk)
    ctx.free(pk)
    assert cnt.value == (steps + 2) * npr, (cnt.value, npr)
    return {"config": "C2 %.0e x %.0e join count(*)" % (npr, nb), "radix_bits": st.radix_bits if st.radix_batches else None, "ms_per_probe_pass": ms,
            "probed_rows_per_s": npr / ms * 1e3, "frac_of_8TBs_at_24B": 24.0 * npr / ms / 1e6 / 8000, "build_kernel_ms": st.build_kernel_ms,
            "partition_ms": st.partition_kernel_ms, "probe_kernel_ms": st.radix_probe_kernel_ms if st.radix_batches else st.probe_kernel_ms}


def agg_c3(ctx, n, groups, vtype, batch):
    """rows are generated batch by batch on the device and pushed device-resident (COL_DEVICE), like a GPU child operator would."""
    lib = ctx.lib
    k, v = ctx.alloc(batch * 8), ctx.alloc(batch * 8)
    cfg = abi.AggCfg()
    cfg.n_group_keys = 1
    cfg.group_key_col[0], cfg.group_key_type[0] = 0, abi.I64
    cfg.n_input_cols = 2
    cfg.input_types[0], cfg.input_types[1] = abi.I64, vtype
    cfg.n_aggs = 3
    for i, (f, col, t) in enumerate([(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, vtype), (abi.AGG_COUNT, -1, abi.I64)]):
        cfg.aggs[i].func, cfg.aggs[i].mode, cfg.aggs[i].arg_col, cfg.aggs[i].arg_type = f, abi.MODE_COMPLETE, col, t
    cfg.est_groups = groups
    h = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    push_ms, gen_ms = 0.0, 0.0
    done = 0
    while done < n:
        m = min(batch, n - done)
        ctx.timer_start()
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=3, col=0, m=groups, start=done), m, k)
        if vtype == abi.F64:
            ctx.gen_column(spec(abi.GEN_RAND_F64, table=3, col=1, start=done), m, v)
        else:
            ctx.gen_column(spec(abi.GEN_RAND_MOD, table=3, col=1, m=1000, start=done), m, v)
        gen_ms += ctx.timer_stop_ms()
        cols = (abi.Col * 2)(dev_col(k, m), dev_col(v, m, vtype))
        ctx.timer_start()
        _lib.check(lib.tsq_agg_push(h, cols, 2, m), h)
        push_ms += ctx.timer_stop_ms()
        done += m
    ctx.timer_start()
    _lib.check(lib.tsq_agg_finish(h), h)
    fin_ms = ctx.timer_stop_ms()
    ng = C.c_int64(0)
    _lib.check(lib.tsq_agg_num_groups(h, C.byref(ng)), h)
    lib.tsq_agg_destroy(h)
    ctx.free(k)
    ctx.free(v)
    algo = 16.0 * n + 24.0 * ng.value
    return {"config": "C3 GROUP BY k: SUM(v %s), COUNT(*) on %.0e rows / %.0e groups" % ("f64" if vtype == abi.F64 else "i64", n, groups), "groups": ng.value,
            "update_ms": push_ms, "finalize_ms": fin_ms, "rows_per_s": n / (push_ms + fin_ms) * 1e3, "algorithmic_GBs": algo / (push_ms + fin_ms) / 1e6,
            "frac_of_8TBs": algo / (push_ms + fin_ms) / 1e6 / 8000, "gen_ms_untimed": gen_ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agg-rows", type=float, default=1e9)
    ap.add_argument("--agg-batch", type=float, default=1e8)
    ap.add_argument("--skip-join", action="store_true")
    ap.add_argument("--skip-agg", action="store_true")
    a = ap.parse_args()
    with _lib.Context(0) as ctx:
        if not a.skip_join:
            for radix in (abi.RADIX_AUTO, abi.RADIX_OFF):
                print(json.dumps(join_c2(ctx, 10_000_000, 100_000_000, 10, radix)), flush=True)
        if not a.skip_agg:
            for vt in (abi.I64, abi.F64):
                print(json.dumps(agg_c3(ctx, int(a.agg_rows), 1_000_000, vt, int(a.agg_batch))), flush=True)


if __name__ == "__main__":
    main()
