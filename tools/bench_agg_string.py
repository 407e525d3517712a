#!/usr/bin/env python3
"""GROUP BY a string key column (16-byte cells): SELECT s, SUM(v), COUNT(*) GROUP BY s over n rows / g groups, device-resident input.
   python tools/bench_agg_string.py [rows [groups [--no-dict]]]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
ctx = _lib.Context(0)
if len(sys.argv) > 3 and sys.argv[3] == "--no-dict":  # the several-column upsert (round 4's route) for comparison
    ctx.set_knob(abi.KNOB_KEYREC, 0)
if len(sys.argv) > 3 and sys.argv[3] == "--arrays":  # records, row ids and travelling cells in separate arrays instead of 64-byte slots
    ctx.set_knob(abi.KNOB_KEYREC, 2)
lib = ctx.lib
rng = np.random.default_rng(5)
k = rng.integers(0, g, n)
v = rng.integers(0, 1000, n)
a = (k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x1234567)
b = (k.astype(np.uint64) + np.uint64(77)) * np.uint64(0xC2B2AE3D27D4EB4F)
sdata = np.ascontiguousarray(np.stack([a, b], axis=1)).view(np.uint8).reshape(-1)
offs = np.arange(n + 1, dtype=np.int64) * 16
dev = []


def up(arr):
    p = ctx.alloc(arr.nbytes + 64)
    ctx.h2d(p, np.ascontiguousarray(arr))
    dev.append(p)
    return p


cols = (abi.Col * 2)()
cols[0].data, cols[0].offsets, cols[0].length, cols[0].elem_size, cols[0].type, cols[0].flags = up(sdata), up(offs), n, -1, abi.BYTES, abi.COL_DEVICE
cols[1].data, cols[1].length, cols[1].elem_size, cols[1].type, cols[1].flags = up(v), n, 8, abi.I64, abi.COL_DEVICE
cfg = abi.AggCfg()
cfg.n_group_keys = 1
cfg.group_key_col[0], cfg.group_key_type[0] = 0, abi.BYTES
cfg.n_input_cols = 2
cfg.input_types[0], cfg.input_types[1] = abi.BYTES, abi.I64
cfg.n_aggs = 2
for i, (f, col) in enumerate([(abi.AGG_SUM, 1), (abi.AGG_COUNT, -1)]):
    cfg.aggs[i].func, cfg.aggs[i].mode, cfg.aggs[i].arg_col, cfg.aggs[i].arg_type = f, abi.MODE_COMPLETE, col, abi.I64
runs = []
for run in range(3):
    h = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    ctx.sync()
    t0 = time.perf_counter()
    _lib.check(lib.tsq_agg_push(h, cols, 2, n), h)
    _lib.check(lib.tsq_agg_finish(h), h)
    ctx.sync()
    runs.append((time.perf_counter() - t0) * 1e3)
    ng = C.c_int64(0)
    _lib.check(lib.tsq_agg_num_groups(h, C.byref(ng)), h)
    lib.tsq_agg_destroy(h)
st = None
print(json.dumps({"mode": sys.argv[3] if len(sys.argv) > 3 else "dict", "rows": n, "groups": int(ng.value), "expected_groups": int(len(np.unique(k))), "ms": min(runs), "runs_ms": runs, "rows_per_s": n / min(runs) * 1e3}))
ctx.close()
