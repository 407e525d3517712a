#!/usr/bin/env python3
"""tsq_rows_encode timing: a partial-aggregate-shaped chunk (int64 count, int64 sum, double sum, int64 key) resident in HBM, encoded
into the RowsData bytes of a coprocessor response in HBM; the oracle appears only in the cpu_baseline leg and in the byte check of a
sample.  NOT YET RUN ON HARDWARE (written after round 1's GPU budget was spent).
usage: bench_encode.py [rows]"""
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


def main():
    import gpu_helpers as G
    from oracle import binding as orc
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
    rng = np.random.default_rng(1)
    chk = Chunk([Column(abi.I64, rng.integers(1, 50, n)), Column(abi.I64, rng.integers(-(1 << 40), 1 << 40, n)), Column(abi.F64, rng.random(n) * 1e5),
                 Column(abi.I64, rng.integers(0, 1 << 28, n))])
    m = min(n, 2_000_000)
    head = Chunk([Column(c.tp, c.data[:m]) for c in chk.columns])
    t = time.perf_counter()
    want_head = orc.encode_rows(head)  # cpu_baseline: the restatement of the EncodeValue loop, one host core
    cpu_s = time.perf_counter() - t
    with _lib.Context(0) as ctx:
        dcols = [G.to_device(ctx, c) for c in chk.columns]
        cap = n * 4 * 11 + 64
        dout, doffs = ctx.alloc(cap), ctx.alloc(8 * (n + 1) + 64)
        try:
            got = C.c_int64(0)
            best = 1e30
            for rep in range(5):
                ctx.sync()
                t = time.perf_counter()
                _lib.check(ctx.lib.tsq_rows_encode(ctx.h, G.dev_cols(dcols), 4, None, n, C.c_void_p(dout), cap, abi.COL_DEVICE, C.c_void_p(doffs), C.byref(got)), ctx.h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
            raw = np.zeros(want_head.size, np.uint8)
            ctx.d2h(raw, dout)
            algo = 8.0 * 4 * n + got.value + 8.0 * n  # values read once + bytes written + row boundaries written
            print(json.dumps({"workload": "encode %d rows x 4 fixed-width columns into EncodeValue response bytes, columns and bytes resident in HBM" % n,
                              "encoded_bytes": int(got.value), "bytes_per_value": got.value / (4.0 * n), "ms": best * 1e3, "values_per_s": 4 * n / best,
                              "algorithmic_GBs": algo / best / 1e9, "frac_of_8TBs": algo / best / 8e12, "head_bytes_equal_oracle": bool((raw == want_head).all()),
                              "cpu_baseline": {"kind": "port", "cores": 1, "values_per_s": 4 * m / cpu_s,
                                               "sample": "oracle restatement of the EncodeValue loop, %d rows x 4 columns, single thread" % m}}))
        finally:
            ctx.free(dout)
            ctx.free(doffs)
            for c in dcols:
                c.free()


if __name__ == "__main__":
    main()
