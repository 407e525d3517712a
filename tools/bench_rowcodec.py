#!/usr/bin/env python3
"""tsq_rowcodec_decode timing: the KV values of a lineitem-shaped table scan (int64 key, int64 day number, double, double, one
more narrow int) in the rowcodec v2 format (generated here with numpy), resident in HBM, decoded into device columns; the oracle
appears only in the cpu_baseline leg.
usage: bench_rowcodec.py [rows]"""
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


def encode_rows_v2(cols, ids):
    """rowcodec.Encoder.Encode (util/rowcodec/encoder.go:34-194) of rows WITHOUT NULLs and with column ids <= 255, vectorised in
    numpy: the INPUT generator of this bench.  cols: int64 / uint64 / float64 arrays; returns (bytes, offsets[n+1]).
    Row = [128][0][n u16][0 u16][ids sorted][end offsets u16][values: ints in 1/2/4/8 little-endian bytes, doubles as 8
    big-endian memcomparable bytes]."""
    order = np.argsort(ids)
    ids = [ids[i] for i in order]
    cols = [cols[i] for i in order]
    assert max(ids) <= 255
    n, k = len(cols[0]), len(cols)
    vals, lens = [], np.zeros((n, k), np.int64)
    for j, c in enumerate(cols):
        if c.dtype == np.float64:
            u = c.view(np.uint64)
            u = np.where(c >= 0, u | np.uint64(1 << 63), ~u)
            vals.append(u.byteswap().view(np.uint8).reshape(n, 8))  # 8 big-endian bytes
            lens[:, j] = 8
        elif c.dtype == np.uint64:
            vals.append(np.ascontiguousarray(c).view(np.uint8).reshape(n, 8))
            lens[:, j] = 1 + (c >= (1 << 8)) + 2 * (c >= (1 << 16)) + 4 * (c >= (1 << 32))
        else:
            v = np.ascontiguousarray(c, dtype=np.int64)
            vals.append(v.view(np.uint8).reshape(n, 8))
            m = v ^ (v >> 63)  # non-negative image: the value fits w bytes iff m < 2^(8w-1)
            lens[:, j] = 1 + (m >= (1 << 7)) + 2 * (m >= (1 << 15)) + 4 * (m >= (1 << 31))
    ends = np.cumsum(lens, axis=1)
    assert int(ends[:, -1].max()) < 65535
    head = np.zeros((n, 6 + 3 * k), np.uint8)
    head[:, 0] = 128
    head[:, 2] = k & 255
    head[:, 3] = k >> 8
    head[:, 6:6 + k] = np.array(ids, np.uint8)[None, :]
    head[:, 6 + k:] = ends.astype("<u2").view(np.uint8).reshape(n, 2 * k)
    full = np.concatenate([head] + vals, axis=1)
    fmask = np.concatenate([np.ones((n, 6 + 3 * k), bool), (np.arange(8)[None, None, :] < lens[:, :, None]).reshape(n, 8 * k)], axis=1)
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(6 + 3 * k + ends[:, -1], out=offsets[1:])
    return full[fmask], offsets


def make_scan(rng, m):
    return [rng.integers(0, 1 << 28, m), rng.integers(0, 2500, m), rng.random(m) * 1e5, rng.integers(0, 11, m) / 100.0, rng.integers(1, 51, m)]


IDS = [1, 2, 3, 4, 5]
TYPES = [abi.I64, abi.I64, abi.F64, abi.F64, abi.I64]


def specs():
    arr = (abi.RowcodecCol * len(IDS))()
    for i, (cid, tp) in enumerate(zip(IDS, TYPES)):
        arr[i].col_id, arr[i].type, arr[i].flags, arr[i].def_bits = cid, tp, 0, 0
    return arr


def cpu_baseline_leg(raw, offs):
    """cpu_baseline: the oracle's restatement of the ChunkDecoder.DecodeToChunk loop (test infrastructure), one host core."""
    from oracle import binding as orc
    t = time.perf_counter()
    st, _ = orc.rowcodec_decode(raw, offs, None, list(zip(IDS, TYPES)))
    assert st == 0
    return time.perf_counter() - t


def main():
    import gpu_helpers as G
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
    rng = np.random.default_rng(1)
    piece = 5_000_000
    raws, offs, base = [], [np.zeros(1, np.int64)], 0
    key_sum = 0
    for lo in range(0, n, piece):
        m = min(piece, n - lo)
        cols = make_scan(rng, m)
        key_sum += int(cols[0].sum())
        b, o = encode_rows_v2(cols, IDS)
        if lo == 0:
            cpu_s, cpu_rows = cpu_baseline_leg(b, o), m
        raws.append(b)
        offs.append(o[1:] + base)
        base += int(o[-1])
    raw, off = np.concatenate(raws), np.concatenate(offs)
    del raws, offs
    with _lib.Context(0) as ctx:
        dbytes, doffs = ctx.alloc(raw.size + 64), ctx.alloc(off.nbytes + 64)
        outs = [G.DevCol(ctx, t, n, with_nulls=True) for t in TYPES]
        try:
            ctx.h2d(dbytes, raw)
            ctx.h2d(doffs, off)
            oc = G.dev_cols(outs)
            m = C.c_int64(0)
            best = 1e30
            for rep in range(5):
                ctx.sync()
                t = time.perf_counter()
                _lib.check(ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes), raw.size, C.c_void_p(doffs), None, n, abi.COL_DEVICE, len(IDS), specs(), oc,
                                                       C.byref(m)), ctx.h)
                ctx.sync()
                best = min(best, time.perf_counter() - t)
            assert m.value == n
            sweep = {}
            if os.environ.get("TSQ_RC_SWEEP"):  # LDS tile size of the kernel (tuning knob of libtsq), best of 3 each
                for kb in (8, 12, 16, 20, 24, 32, 48, 64):
                    ctx.set_knob(abi.KNOB_ROWCODEC_LDS_KB, kb)
                    b3 = 1e30
                    for rep in range(3):
                        ctx.sync()
                        t = time.perf_counter()
                        _lib.check(ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes), raw.size, C.c_void_p(doffs), None, n, abi.COL_DEVICE, len(IDS), specs(),
                                                               oc, C.byref(m)), ctx.h)
                        ctx.sync()
                        b3 = min(b3, time.perf_counter() - t)
                    sweep["%dKB" % kb] = round(b3 * 1e3, 4)
                ctx.set_knob(abi.KNOB_ROWCODEC_LDS_KB)
                ctx.set_knob(abi.KNOB_ROWCODEC_FAST_LAYOUT, 0)  # every wave through the general per-row column search
                b3 = 1e30
                for rep in range(3):
                    ctx.sync()
                    t = time.perf_counter()
                    _lib.check(ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes), raw.size, C.c_void_p(doffs), None, n, abi.COL_DEVICE, len(IDS), specs(), oc,
                                                           C.byref(m)), ctx.h)
                    ctx.sync()
                    b3 = min(b3, time.perf_counter() - t)
                sweep["general_path_only"] = round(b3 * 1e3, 4)
                ctx.set_knob(abi.KNOB_ROWCODEC_FAST_LAYOUT)
                ctx.set_knob(abi.KNOB_ROWCODEC_PIPELINE, 0)  # the plain kernel: offsets -> bytes -> parse per tile, nothing prefetched
                b3 = 1e30
                for rep in range(3):
                    ctx.sync()
                    t = time.perf_counter()
                    _lib.check(ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes), raw.size, C.c_void_p(doffs), None, n, abi.COL_DEVICE, len(IDS), specs(), oc,
                                                           C.byref(m)), ctx.h)
                    ctx.sync()
                    b3 = min(b3, time.perf_counter() - t)
                sweep["not_pipelined"] = round(b3 * 1e3, 4)
                ctx.set_knob(abi.KNOB_ROWCODEC_PIPELINE)
            key = outs[0].to_host().data
            algo = raw.size + 8.0 * n + 8.0 * len(IDS) * n  # row bytes + one 8-byte row boundary + 8 B per decoded value
            print(json.dumps({"workload": "decode %d stored rows (rowcodec v2, %d fixed-width columns), bytes and columns resident in HBM" % (n, len(IDS)),
                              "encoded_bytes": int(raw.size), "bytes_per_row": raw.size / float(n), "ms": best * 1e3, "rows_per_s": n / best,
                              "values_per_s": len(IDS) * n / best, "algorithmic_GBs": algo / best / 1e9, "frac_of_8TBs": algo / best / 8e12,
                              "key_checksum_ok": bool(int(key.astype(np.int64).sum()) == key_sum), "lds_tile_sweep_ms": sweep,
                              "cpu_baseline": {"kind": "port", "cores": 1, "rows_per_s": cpu_rows / cpu_s,
                                               "sample": "oracle restatement of the ChunkDecoder.DecodeToChunk loop, %d rows x %d columns, single thread" % (cpu_rows, len(IDS))}}))
        finally:
            ctx.free(dbytes)
            ctx.free(doffs)
            for o in outs:
                o.free()


if __name__ == "__main__":
    main()
