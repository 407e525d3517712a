#!/usr/bin/env python3
"""tsq_rows_decode timing: a lineitem-shaped response (int64 key, int64 day number, double, double) in the EncodeValue format
(generated here with numpy), resident in HBM, decoded into device columns; the oracle appears only in the cpu_baseline leg.
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
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
import gpu_helpers as G  # noqa: E402


def encode_value_rows(cols):
    """EncodeValue (util/codec/codec.go:74-99,205-209) of rows of int64 / float64 columns, vectorised in numpy: the INPUT
    generator of this bench (flag 8 + zig-zag varint for ints, flag 5 + 8 big-endian memcomparable bytes for doubles)."""
    n = len(cols[0])
    parts, lens = [], []
    for c in cols:
        if c.dtype == np.float64:
            u = c.view(np.uint64)
            u = np.where(c >= 0, u | np.uint64(1 << 63), ~u)
            b = np.zeros((n, 11), np.uint8)
            b[:, 0] = 5
            b[:, 1:9] = u.astype(">u8").view(np.uint8).reshape(n, 8)
            ln = np.full(n, 9, np.int64)
        else:
            v = c.astype(np.int64)
            z = ((v << 1) ^ (v >> 63)).view(np.uint64)          # zig-zag (encoding/binary.PutVarint)
            nb = np.ones(n, np.int64)                             # bytes of the varint: one per started 7-bit group
            for k in range(1, 10):
                nb += ((z >> np.uint64(7 * k)) != 0).astype(np.int64)
            b = np.zeros((n, 11), np.uint8)
            b[:, 0] = 8
            for k in range(10):
                b[:, 1 + k] = ((z >> np.uint64(7 * k)) & np.uint64(0x7F)).astype(np.uint8) | ((nb > k + 1).astype(np.uint8) << 7)
            ln = nb + 1                                           # + flag byte
        parts.append(b)
        lens.append(ln)
    # interleave column-wise per row, dropping the unused tail bytes of every value
    allb = np.stack(parts, axis=1).reshape(n * len(cols), 11)
    alll = np.stack(lens, axis=1).reshape(n * len(cols))
    mask = np.arange(11)[None, :] < alll[:, None]
    return allb[mask]


def cpu_baseline_leg(raw, types, rows):
    """cpu_baseline: the oracle's restatement of readRowsData + DecodeOne (test infrastructure), timed on one host core."""
    from oracle import binding as orc
    t = time.perf_counter()
    st, _, _ = orc.decode_rows(raw, types, rows)
    assert st == 0
    return time.perf_counter() - t


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
    rng = np.random.default_rng(1)
    types = [abi.I64, abi.I64, abi.F64, abi.F64]
    piece = 5_000_000
    raws = []
    for lo in range(0, n, piece):
        m = min(piece, n - lo)
        raws.append(encode_value_rows([rng.integers(0, 1 << 28, m), rng.integers(0, 2500, m), rng.random(m) * 1e5, rng.integers(0, 11, m) / 100.0]))
        if lo == 0:
            cpu_s = cpu_baseline_leg(raws[0], types, m)
            cpu_vals = m * 4
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
            # the same response through the chunk-parallel decoder (one lane per 64-row tipb.Chunk; the route for var-len schemas)
            chunk_best = None
            if len(sys.argv) > 2 and sys.argv[2] == "--chunks":
                # chunk boundaries every 64 rows: positions of the rows are known to the generator (4 values per row)
                # the chunk boundaries: the generator's own rows walked once on the host (input preparation, untimed)
                pos, offs = 0, [0]
                view = raw
                for lo in range(0, n, 64):
                    hi = min(lo + 64, n)
                    for _ in range((hi - lo) * 4):
                        f = view[pos]
                        if f == 5:
                            pos += 9
                        else:
                            pos += 1
                            while view[pos] & 0x80:
                                pos += 1
                            pos += 1
                    offs.append(pos)
                offs = np.array(offs, np.int64)
                doffs = ctx.alloc(offs.nbytes + 64)
                ctx.h2d(doffs, offs)
                chunk_best = 1e30
                for rep in range(5):
                    ctx.sync()
                    t = time.perf_counter()
                    _lib.check(ctx.lib.tsq_rows_decode_chunks(ctx.h, C.c_void_p(dbytes), raw.size, C.c_void_p(doffs), len(offs) - 1, abi.COL_DEVICE, 4, tp, oc, n, C.byref(m)), ctx.h)
                    ctx.sync()
                    chunk_best = min(chunk_best, time.perf_counter() - t)
                assert m.value == n
                ctx.free(doffs)
            key = outs[0].to_host().data
            algo = raw.size + 8.0 * 4 * n
            print(json.dumps({"workload": "decode %d rows x 4 fixed-width columns of an EncodeValue response, bytes and columns resident in HBM" % n,
                              "encoded_bytes": int(raw.size), "bytes_per_value": raw.size / (4.0 * n), "ms": best * 1e3, "values_per_s": 4 * n / best,
                              "input_GBs": raw.size / best / 1e9, "algorithmic_GBs": algo / best / 1e9, "frac_of_8TBs": algo / best / 8e12,
                              "key_checksum_ok": bool(int(key.sum()) == int(key.astype(np.int64).sum())),
                              "chunks_route": None if chunk_best is None else {"ms": chunk_best * 1e3, "values_per_s": 4 * n / chunk_best, "frac_of_8TBs": algo / chunk_best / 8e12,
                                                                                "note": "tsq_rows_decode_chunks: one lane per 64-row chunk"},
                              "cpu_baseline": {"kind": "port", "cores": 1, "values_per_s": cpu_vals / cpu_s,
                                               "sample": "oracle restatement of readRowsData + DecodeOne, %d rows x 4 columns, single thread" % (cpu_vals // 4)}}))
        finally:
            ctx.free(dbytes)
            for o in outs:
                o.free()


if __name__ == "__main__":
    main()
