#!/usr/bin/env python3
"""tsq_rows_decode timing: a lineitem-shaped response (int64 key, int64 day number, double, double) encoded with EncodeValue by the
oracle (test infrastructure: generator + CPU baseline only), resident in HBM, decoded into device columns.
usage: bench_decode.py [rows]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as orc  # noqa: E402  (generator + cpu baseline leg)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
from tinysql_amd.chunk import Chunk, Column  # noqa: E402
import gpu_helpers as G  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
    rng = np.random.default_rng(1)
    types = [abi.I64, abi.I64, abi.F64, abi.F64]
    piece = 5_000_000
    raws = []
    for lo in range(0, n, piece):
        m = min(piece, n - lo)
        chk = Chunk([Column(abi.I64, rng.integers(0, 1 << 28, m)), Column(abi.I64, rng.integers(0, 2500, m)), Column(abi.F64, rng.random(m) * 1e5),
                     Column(abi.F64, rng.integers(0, 11, m) / 100.0)])
        raws.append(orc.encode_rows(chk))
        if lo == 0:
            t = time.perf_counter()
            st, _, _ = orc.decode_rows(raws[0], types, m)
            cpu_s = time.perf_counter() - t
            cpu_vals = m * 4
            assert st == 0
    raw = np.concatenate(raws)
    del raws
    with _lib.Context(0) as ctx:
        dbytes = ctx.alloc(raw.size + 64)
        outs = [G.DevCol(ctx, t, n, with_nulls=True) for t in types]
        try:
            ctx.h2d(dbytes, raw)
            oc = G.dev_cols(outs)
            tp = (C.c_int32 * 4)(*types)
            m, used = C.c_int64(0), C.c_int64(0)
            best = 1e30
            for rep in range(5):
                ctx.sync()
                t = time.perf_counter()
                _lib.check(ctx.lib.tsq_rows_decode(ctx.h, C.c_void_p(dbytes), raw.size, abi.COL_DEVICE, 4, tp, oc, n, C.byref(m), C.byref(used)), ctx.h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
            assert m.value == n and used.value == raw.size
            key = outs[0].to_host().data
            algo = raw.size + 8.0 * 4 * n
            print(json.dumps({"workload": "decode %d rows x 4 fixed-width columns of an EncodeValue response, bytes and columns resident in HBM" % n,
                              "encoded_bytes": int(raw.size), "bytes_per_value": raw.size / (4.0 * n), "ms": best * 1e3, "values_per_s": 4 * n / best,
                              "input_GBs": raw.size / best / 1e9, "algorithmic_GBs": algo / best / 1e9, "frac_of_8TBs": algo / best / 8e12,
                              "key_checksum_ok": bool(int(key.sum()) == int(key.astype(np.int64).sum())),
                              "cpu_baseline": {"kind": "port", "cores": 1, "values_per_s": cpu_vals / cpu_s,
                                               "sample": "oracle restatement of readRowsData + DecodeOne, %d rows x 4 columns, single thread" % (cpu_vals // 4)}}))
        finally:
            ctx.free(dbytes)
            for o in outs:
                o.free()


if __name__ == "__main__":
    main()
