#!/usr/bin/env python3
"""TPC-H Q3-shaped query (BASELINE.json configs[4]) on device-resident operators:

    SELECT l_orderkey, o_orderdate, o_shippriority, SUM(l_extendedprice * (1 - l_discount)) AS revenue
    FROM customer, orders, lineitem
    WHERE c_mktsegment = SEG AND c_custkey = o_custkey AND l_orderkey = o_orderkey AND o_orderdate < D AND l_shipdate > D
    GROUP BY l_orderkey, o_orderdate, o_shippriority   [ORDER BY revenue DESC, o_orderdate LIMIT 10 with --topn]

TinySQL has int / real / string types only (types/eval_type.go:21-28): dates are day numbers, the market segment an int code.
Plan (what planner/core would produce with hash joins): Selection(customer) -> build;  Selection(orders) probes it;
that result is the build side for Selection(lineitem);  Projection(revenue);  HashAgg.
usage: q3.py [SF]   (customer 1.5e5*SF, orders 1.5e6*SF, lineitem 6e6*SF rows)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
from tinysql_amd import expression as E  # noqa: E402
from tinysql_amd import gpu_pipeline as G  # noqa: E402
from tinysql_amd.chunk import Chunk, Column, StrColumn  # noqa: E402
from tinysql_amd.executor import AggFuncDesc  # noqa: E402

SEG, D = 1, 1200
I, R = abi.I64, abi.F64


SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]  # TPC-H c_mktsegment; SEGMENTS[SEG] = the query's


def tables(sf, seed=7, string_segment=False):
    """host-side synthetic tables (numpy): deterministic, FK-consistent.  string_segment: c_mktsegment as the varchar it is in
    TPC-H (the filter is then EQString, builtin_compare_vec_generated.go:393) instead of an int code."""
    rng = np.random.default_rng(seed)
    nc, no, nl = int(150_000 * sf), int(1_500_000 * sf), int(6_000_000 * sf)
    seg = rng.integers(0, 5, nc)
    customer = Chunk([Column(I, np.arange(nc, dtype=np.int64)), StrColumn([SEGMENTS[i] for i in seg.tolist()]) if string_segment else Column(I, seg)])  # c_custkey, c_mktsegment
    orders = Chunk([Column(I, rng.permutation(no).astype(np.int64)), Column(I, rng.integers(0, nc, no)),               # o_orderkey, o_custkey
                    Column(I, rng.integers(0, 2400, no)), Column(I, rng.integers(0, 3, no))])                            # o_orderdate, o_shippriority
    lineitem = Chunk([Column(I, rng.integers(0, no, nl)), Column(I, rng.integers(0, 2500, nl)),                         # l_orderkey, l_shipdate
                      Column(R, rng.random(nl) * 1e5), Column(R, rng.integers(0, 11, nl) / 100.0)])                      # l_extendedprice, l_discount
    return customer, orders, lineitem


def tables_device(ctx, sf, seed=7):
    """the same shapes generated in HBM (tsq_gen_column: counter-based splitmix64) — SF 100 is 24 GB of
    columns that never cross PCIe.  o_orderkey is a bijection of [0, N_orders); prices / discounts are uniform in [0, 1)."""
    nc, no, nl = int(150_000 * sf), int(1_500_000 * sf), int(6_000_000 * sf)

    def spec(kind, table, col, **kw):
        g = abi.GenSpec()
        g.kind, g.table, g.col, g.seed = kind, table, col, seed
        for k, v in kw.items():
            setattr(g, k, v)
        return g

    def table(n, specs, types):
        cols = []
        for sp, tp in zip(specs, types):
            c = G.DeviceColumn(ctx, tp, n)
            ctx.gen_column(sp, n, c.data)
            ctx.memset(c.bitmap, 0xFF, (n + 7) // 8)
            cols.append(c)
        return G.DeviceChunk(cols, n)

    a = 2654435761  # prime: (a * i + b) mod N is a bijection of [0, N) for every N it does not divide
    customer = table(nc, [spec(abi.GEN_SEQ, 1, 0), spec(abi.GEN_RAND_MOD, 1, 1, m=5)], [I, I])
    orders = table(no, [spec(abi.GEN_AFFINE, 2, 0, a=a, b=12345, m=no), spec(abi.GEN_RAND_MOD, 2, 1, m=nc), spec(abi.GEN_RAND_MOD, 2, 2, m=2400),
                        spec(abi.GEN_RAND_MOD, 2, 3, m=3)], [I, I, I, I])
    lineitem = table(nl, [spec(abi.GEN_RAND_MOD, 3, 0, m=no), spec(abi.GEN_RAND_MOD, 3, 1, m=2500), spec(abi.GEN_RAND_F64, 3, 2), spec(abi.GEN_RAND_F64, 3, 3)],
                     [I, I, R, R])
    return customer, orders, lineitem


JOINS = []  # the two join operators of the last plan (their route statistics are reported)
ROUTES = {0: "direct", 1: "radix, slices through L2", 2: "radix, 64-bit LDS images", 3: "packed keys"}


def plan(ctx, customer_d, orders_d, lineitem_d, batch_rows=1 << 24, jit=None, topn=0, string_segment=False):
    F, Col, K = E.ScalarFunction, E.Column, E.Constant
    if string_segment:  # WHERE c_mktsegment = 'BUILDING' on the varchar column; the join below only needs c_custkey (column pruning)
        sel = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, customer_d, batch_rows), [F("eq", Col(1, abi.BYTES), K(SEGMENTS[SEG]))], jit=jit)
        cust = G.GpuProjectionExec(ctx, sel, [Col(0, I), Col(0, I)], jit=jit)  # (two columns so that the join's output keeps its column numbers)
    else:
        cust = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, customer_d, batch_rows), [F("eq", Col(1, I), K(SEG))], jit=jit)
    ords = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, orders_d, batch_rows), [F("lt", Col(2, I), K(D))], jit=jit)
    # orders (probe, left) JOIN customer (build, right) ON o_custkey = c_custkey  ->  o_orderkey,o_custkey,o_orderdate,o_shippriority,c_custkey,c_mktsegment
    j1 = G.GpuHashJoinExec(ctx, ords, cust, [1], [0], abi.JOIN_INNER, 1)
    JOINS[:] = [j1]
    line = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, lineitem_d, batch_rows), [F("gt", Col(1, I), K(D))], jit=jit)
    # lineitem (probe, left) JOIN j1 (build, right) ON l_orderkey = o_orderkey -> l_orderkey,l_shipdate,price,disc | o_orderkey,o_custkey,o_orderdate,o_shippriority,c_*,c_*
    j2 = G.GpuHashJoinExec(ctx, line, j1, [0], [0], abi.JOIN_INNER, 1)
    JOINS.append(j2)
    proj = G.GpuProjectionExec(ctx, j2, [Col(0, I), Col(6, I), Col(7, I), F("mul", Col(2, R), F("minus", K(1.0), Col(3, R)))], jit=jit)
    aggs = [AggFuncDesc(abi.AGG_FIRSTROW, 0, I), AggFuncDesc(abi.AGG_FIRSTROW, 1, I), AggFuncDesc(abi.AGG_FIRSTROW, 2, I), AggFuncDesc(abi.AGG_SUM, 3, R)]
    agg = G.GpuHashAggExec(ctx, proj, [0, 1, 2], aggs)
    if not topn:
        return agg
    # ... ORDER BY revenue DESC, o_orderdate LIMIT topn (TopNExec, executor/sort.go:146-318); agg output: orderkey, date, prio, revenue
    return G.GpuSortExec(ctx, agg, [3, 1], [True, False], offset=0, count=topn, pull_rows=max(8, topn))


def reference(customer, orders, lineitem):
    """plain numpy restatement of the query (checker for the small test)."""
    ck = customer.columns[0].data
    cs = customer.columns[1].data if customer.columns[1].tp != abi.BYTES else np.array([SEGMENTS.index(v) for v in customer.columns[1].values()])
    ok, oc, od, op = (c.data for c in orders.columns)
    lk, ls, lp, ld = (c.data for c in lineitem.columns)
    good_c = np.zeros(len(ck), bool)
    good_c[ck[cs == SEG]] = True
    o_sel = (od < D) & good_c[oc]
    date_of = np.full(len(ok), -1, np.int64)
    prio_of = np.zeros(len(ok), np.int64)
    date_of[ok[o_sel]] = od[o_sel]
    prio_of[ok[o_sel]] = op[o_sel]
    l_sel = (ls > D) & (date_of[lk] >= 0)
    keys = lk[l_sel]
    rev = lp[l_sel] * (1.0 - ld[l_sel])
    order = np.argsort(keys, kind="stable")
    keys, rev = keys[order], rev[order]
    uk, start = np.unique(keys, return_index=True)
    sums = np.add.reduceat(rev, start) if len(keys) else np.zeros(0)
    return uk, date_of[uk], prio_of[uk], sums


class TimedLib:
    """--trace: wall time per C-ABI entry point (host view: launch + synchronisation + ctypes marshalling)."""

    def __init__(self, lib):
        self._lib, self.acc = lib, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def wrapped(*a):
            t = time.perf_counter()
            r = fn(*a)
            e = self.acc.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t
            return r
        return wrapped


def main():
    trace = "--trace" in sys.argv
    if trace:
        sys.argv.remove("--trace")
    topn = 10 if "--topn" in sys.argv else 0
    if topn:
        sys.argv.remove("--topn")
    on_device = "--device-gen" in sys.argv
    if on_device:
        sys.argv.remove("--device-gen")
    strseg = "--string-segment" in sys.argv
    if strseg:
        sys.argv.remove("--string-segment")
        assert not on_device, "--string-segment builds the customer table on the host"
    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    t0 = time.time()
    if not on_device:
        customer, orders, lineitem = tables(sf, string_segment=strseg)
    gen_s = time.time() - t0
    with _lib.Context(0) as ctx:
        if on_device:
            dev = list(tables_device(ctx, sf))
            ctx.sync()
            gen_s = time.time() - t0
            customer, orders, lineitem = dev
        else:
            dev = [G.DeviceChunk.from_host(ctx, t) for t in (customer, orders, lineitem)]
        try:
            best, best_exec, groups = 1e30, 1e30, 0
            for rep in range(4):
                if trace and rep == 3:
                    ctx.lib = TimedLib(ctx.lib)
                exe = plan(ctx, *dev, topn=topn, string_segment=strseg)
                ctx.sync()
                t1 = time.perf_counter()
                exe.Open()
                t_exec, out = 0.0, []
                try:
                    while True:
                        t2 = time.perf_counter()
                        chk = exe.Next()   # synchronous: the chunk is complete in HBM when Next returns
                        t_exec += time.perf_counter() - t2
                        if chk.NumRows() == 0:
                            break
                        out.append(chk.to_host())
                finally:
                    exe.Close()
                ctx.sync()
                dt = time.perf_counter() - t1
                if dt < best:
                    best, best_exec = dt, t_exec
                groups = sum(c.NumRows() for c in out)
            if trace:
                tl, ctx.lib = ctx.lib, ctx.lib._lib
                for k, (n, t) in sorted(tl.acc.items(), key=lambda kv: -kv[1][1]):
                    print("%-28s %5d calls %9.3f ms" % (k, n, t * 1e3), file=sys.stderr)
                print("last rep: total %.3f ms, in Next %.3f ms" % (dt * 1e3, t_exec * 1e3), file=sys.stderr)
            rows_in = customer.NumRows() + orders.NumRows() + lineitem.NumRows()
            print(json.dumps({"query": "TPC-H Q3-shaped, device-resident Selection->Join->Join->Projection->HashAgg" + ("->TopN(10)" if topn else "") + (", c_mktsegment = 'BUILDING' on a varchar column" if strseg else ""), "SF": sf, "input_rows": rows_in,
                              "tables": "generated in HBM (tsq_gen_column)" if on_device else "numpy, copied to HBM once",
                              "groups": groups, "best_s": best, "exec_s_result_in_hbm": best_exec, "input_rows_per_s": rows_in / best,
                              "input_rows_per_s_result_in_hbm": rows_in / best_exec, "host_table_gen_s": gen_s,
                              "joins": [{"build_rows": int(j.last_stats.build_rows), "probe_rows": int(j.last_stats.probe_rows), "out_rows": int(j.last_stats.out_rows),
                                         "route_of_last_batch": ROUTES.get(j.last_stats.probe_route, "?"), "radix_batches": int(j.last_stats.radix_batches),
                                         "packed_key_bits": int(j.last_stats.packed_key_bits)} for j in JOINS if getattr(j, "last_stats", None) is not None]}))
        finally:
            for d in dev:
                d.free()


if __name__ == "__main__":
    main()
