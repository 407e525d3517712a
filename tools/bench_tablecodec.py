#!/usr/bin/env python3
"""tsq_rowkeys_encode / tsq_rowkeys_decode timing: the record keys of a table scan resident in HBM.
usage: bench_tablecodec.py [keys]"""
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


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    with _lib.Context(0) as ctx:
        lib = ctx.lib
        dh, dk, dh2 = ctx.alloc(8 * n + 64), ctx.alloc(19 * n + 64), ctx.alloc(8 * n + 64)
        s = abi.GenSpec()
        s.kind, s.seed, s.table, s.col, s.m = abi.GEN_RAND_MOD, 7, 3, 0, 1 << 62
        ctx.gen_column(s, n, dh)
        ctx.sync()
        out = {"keys": n}
        got = C.c_int64(0)
        for name, call in (("encode", lambda: lib.tsq_rowkeys_encode(ctx.h, 41, C.c_void_p(dh), n, abi.COL_DEVICE, C.c_void_p(dk))),
                           ("decode", lambda: lib.tsq_rowkeys_decode(ctx.h, C.c_void_p(dk), 19 * n, None, n, abi.COL_DEVICE, C.c_void_p(dh2), None, C.byref(got)))):
            _lib.check(call(), ctx.h)
            ctx.sync()
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                _lib.check(call(), ctx.h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
            out[name + "_ms"] = best * 1e3
            out[name + "_frac"] = 27.0 * n / best / 8e12  # 19 B key + 8 B handle per key against the 8 TB/s peak
        # round trip: the decoded handles are the generated ones
        a, b = np.zeros(1 << 20, np.int64), np.zeros(1 << 20, np.int64)
        ctx.d2h(a, dh)
        ctx.d2h(b, dh2)
        out["verified"] = bool(got.value == n and (a == b).all())
        print(json.dumps(out))


if __name__ == "__main__":
    main()
