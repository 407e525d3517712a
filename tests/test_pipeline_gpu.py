"""GPU: device-resident operator pipeline (tinysql_amd/gpu_pipeline.py) on a Q3-shaped plan —
Selection -> HashJoin -> HashJoin -> Projection -> HashAgg with chunks that never leave HBM — against a plain numpy
restatement of the query, plus tsq_chunk_compact on its own against numpy boolean indexing."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import gpu_pipeline as GP
from tinysql_amd.chunk import Chunk, Column

from . import helpers as H

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import q3  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 70001])
def test_chunk_compact_matches_boolean_indexing(ctx, n):
    rng = np.random.default_rng(n)
    chk = Chunk([Column(abi.I64, rng.integers(-9, 9, n), rng.random(n) > 0.2), Column(abi.F64, rng.random(n)),
                 Column(abi.F32, rng.random(n).astype(np.float32), rng.random(n) > 0.5), Column(abi.U64, rng.integers(0, 1 << 62, n).astype(np.uint64))])
    sel = rng.random(n) > 0.4
    dev = GP.DeviceChunk.from_host(ctx, chk)
    out = [GP.DeviceColumn(ctx, c.tp, n) for c in chk.columns]
    flags = ctx.alloc(n + 64)
    try:
        ctx.h2d(flags, sel.astype(np.uint8))
        m = C.c_int64(0)
        oc = (abi.Col * 4)(*[c.col(n) for c in out])
        _lib.check(ctx.lib.tsq_chunk_compact(ctx.h, dev.cols(), 4, n, flags, oc, C.byref(m)), ctx.h)
        assert m.value == int(sel.sum())
        got = GP.DeviceChunk(out, m.value).to_host()
        want = Chunk([Column(c.tp, c.data[sel], None if c.notnull is None else c.notnull[sel]) for c in chk.columns])
        assert H.rows_equal_unordered(got, want)
    finally:
        dev.free()
        for c in out:
            c.free()
        ctx.free(flags)


@pytest.mark.parametrize("jit", [abi.JIT_OFF, abi.JIT_FORCE])
def test_q3_shaped_plan_on_device_chunks(ctx, jit):
    customer, orders, lineitem = q3.tables(0.05)  # 7.5e3 / 7.5e4 / 3e5 rows
    dev = [GP.DeviceChunk.from_host(ctx, t) for t in (customer, orders, lineitem)]
    try:
        out = GP.drain_device(q3.plan(ctx, *dev, batch_rows=50_000, jit=jit))  # several batches per table
        uk, dates, prios, sums = q3.reference(customer, orders, lineitem)
        got = {}
        for c in out:
            for k, d, p, s in c.rows():
                assert k not in got
                got[k] = (d, p, s)
        assert len(got) == len(uk) > 1000
        for k, d, p, s in zip(uk.tolist(), dates.tolist(), prios.tolist(), sums.tolist()):
            g = got[k]
            assert g[0] == d and g[1] == p
            assert abs(g[2] - s) <= 1e-9 * max(1.0, abs(s))  # double SUM re-ordering bound, a handful of rows per group
    finally:
        for d in dev:
            d.free()
