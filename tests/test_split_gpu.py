"""GPU: hash-radix redistribute (tsq_radix_split) — every row lands in exactly the part its key
ranks to, nothing is lost or duplicated, NULL keys go to part 0."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rank_fn():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    lib.sim_key_rank.restype = C.c_uint32
    lib.sim_key_rank.argtypes = [C.c_uint64, C.c_uint32]
    return lib.sim_key_rank


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_radix_split_partitions_exactly(ctx, parts):
    rank = _rank_fn()
    rng = np.random.default_rng(40 + parts)
    n = 50_001
    key = Column(abi.I64, rng.integers(-1000, 1000, n), rng.random(n) > 0.03)
    pay = H.random_column(rng, abi.F64, n, 0.2)
    f32 = Column(abi.F32, rng.random(n).astype(np.float32))
    src = [G.to_device(ctx, c) for c in (key, pay, f32)]
    dst = [G.DevCol(ctx, c.tp, n, with_nulls=c.notnull is not None) for c in (key, pay, f32)]
    try:
        counts = (C.c_int64 * parts)()
        _lib.check(ctx.lib.tsq_radix_split(ctx.h, G.dev_cols(src), 3, 0, 0, n, parts, G.dev_cols(dst), counts), ctx.h)
        counts = list(counts)
        assert sum(counts) == n
        out = Chunk([d.to_host() for d in dst])
        rows = out.rows()
        off = 0
        for p in range(parts):
            for r in rows[off:off + counts[p]]:
                want = 0 if r[0] is None else rank(r[0] & ((1 << 64) - 1), parts)
                assert want == p
            off += counts[p]
        assert H.rows_equal_unordered(rows, Chunk([key, pay, f32]).rows())
    finally:
        for d in src + dst:
            d.free()


@pytest.mark.parametrize("parts,ncols", [(2, 1), (4, 2), (8, 3), (8, 1)])
def test_radix_split_fast_path_tile_sort(ctx, parts, ncols):
    # 8-byte columns without NULL bitmaps and >= 64 Ki rows take the LDS tile-sort path (k_rank_hist +
    # k_radix_partition with exact region bases): same contract — contiguous runs, every row in the part its key ranks to,
    # payload cells travel with their key.
    rank = _rank_fn()
    rng = np.random.default_rng(7 * parts + ncols)
    n = 300_007
    k = rng.integers(-10**12, 10**12, n)
    cols = [Column(abi.I64, k)] + [Column(abi.I64, k * (j + 2) + j) for j in range(ncols - 1)]
    src = [G.to_device(ctx, c) for c in cols]
    dst = [G.DevCol(ctx, abi.I64, n) for _ in cols]
    try:
        counts = (C.c_int64 * parts)()
        _lib.check(ctx.lib.tsq_radix_split(ctx.h, G.dev_cols(src), ncols, 0, 0, n, parts, G.dev_cols(dst), counts), ctx.h)
        counts = list(counts)
        assert sum(counts) == n
        out = [d.to_host().data for d in dst]
        want_rank = np.array([rank(int(x) & ((1 << 64) - 1), parts) for x in out[0][:: max(1, n // 5000)]])
        bounds = np.cumsum([0] + counts)
        got_part = np.searchsorted(bounds, np.arange(n)[:: max(1, n // 5000)], side="right") - 1
        assert np.array_equal(want_rank, got_part)
        for j in range(1, ncols):
            assert np.array_equal(out[j], out[0] * (j + 1) + (j - 1))
        assert np.array_equal(np.sort(out[0]), np.sort(k))
    finally:
        for d in src + dst:
            d.free()


@pytest.mark.parametrize("parts,long_cells", [(1, False), (3, False), (8, False), (4, True)])
def test_radix_split_carries_var_len_payload_columns(ctx, parts, long_cells):
    # a var-len column is payload of the split: its cells follow their rows (lengths -> scan = the offsets of the split order, then
    # the bytes), NULL cells have no bytes; rows inside a part keep no particular order, so the parts are compared as multisets
    from tinysql_amd.chunk import StrColumn
    rank = _rank_fn()
    rng = np.random.default_rng(70 + parts)
    n = 2000 if long_cells else 40_003
    key = Column(abi.I64, rng.integers(-500, 500, n), rng.random(n) > 0.03)
    width = 3000 if long_cells else 24
    names = StrColumn([None if rng.random() < 0.1 else (b"" if rng.random() < 0.1 else bytes(rng.integers(65, 91, int(rng.integers(1, width)), dtype=np.uint8))) for _ in range(n)])
    tags = StrColumn([b"t%d" % (i % 97) for i in range(n)])
    pay = H.random_column(rng, abi.F64, n, 0.2)
    src = [G.to_device(ctx, key), G.DevStrCol(ctx, names), G.to_device(ctx, pay), G.DevStrCol(ctx, tags)]
    dst = [G.DevCol(ctx, abi.I64, n, True), G.DevStrCol(ctx, nrows=n, nbytes=len(names.data)), G.DevCol(ctx, abi.F64, n, True),
           G.DevStrCol(ctx, nrows=n, nbytes=len(tags.data))]
    try:
        counts = (C.c_int64 * parts)()
        _lib.check(ctx.lib.tsq_radix_split(ctx.h, G.dev_cols(src), 4, 0, 0, n, parts, G.dev_cols(dst), counts), ctx.h)
        counts = list(counts)
        assert sum(counts) == n
        out = Chunk([dst[0].to_host(), dst[1].to_host(n, len(names.data)), dst[2].to_host(), dst[3].to_host(n, len(tags.data))])
        rows, want = out.rows(), Chunk([key, names, pay, tags]).rows()
        off = 0
        for p in range(parts):
            for r in rows[off:off + counts[p]]:
                assert (0 if r[0] is None else rank(r[0] & ((1 << 64) - 1), parts)) == p
            off += counts[p]
        assert H.rows_equal_unordered(rows, want)
        # a var-len KEY column: equal strings land in the same part, NULL strings in part 0
        counts2 = (C.c_int64 * parts)()
        _lib.check(ctx.lib.tsq_radix_split(ctx.h, G.dev_cols(src), 4, 3, 0, n, parts, G.dev_cols(dst), counts2), ctx.h)
        counts2 = list(counts2)
        out = Chunk([dst[0].to_host(), dst[1].to_host(n, len(names.data)), dst[2].to_host(), dst[3].to_host(n, len(tags.data))])
        rows = out.rows()
        assert sum(counts2) == n and H.rows_equal_unordered(rows, want)
        part_of, off = {}, 0
        for p in range(parts):
            for r in rows[off:off + counts2[p]]:
                assert part_of.setdefault(r[3], p) == p
            off += counts2[p]
        if parts > 1:
            assert len(set(part_of.values())) > 1  # 97 distinct tags do not all rank to one part
    finally:
        for d in src + dst:
            d.free()
