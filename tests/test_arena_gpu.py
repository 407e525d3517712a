"""GPU: operators on a context with an ARENA (tsq_ctx_reserve): every buffer they need is carved from one slab reserved up front —
same results as the oracle, nothing left in the arena once the operators are gone, a too-small arena falls through to hipMalloc."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import out_types_for

pytestmark = pytest.mark.gpu


def _join_and_agg(ctx, orc):
    rng = np.random.default_rng(4)
    t = [abi.I64, abi.I64]
    build = Chunk([Column(abi.I64, rng.integers(0, 5000, 20_000)), Column(abi.I64, rng.integers(0, 99, 20_000))])
    probe = Chunk([Column(abi.I64, rng.integers(0, 6000, 70_000), rng.random(70_000) > 0.02), Column(abi.I64, np.arange(70_000))])
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_LEFT_OUTER, 1)
    assert H.rows_equal_unordered(G.run_join(ctx, cfg, build, probe, chunk_rows=4096), orc.hash_join(cfg, build, probe))
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    acfg = H.agg_cfg(t, [0], aggs)
    assert H.rows_equal_unordered(G.run_agg(ctx, acfg, probe, out_types_for(aggs), chunk_rows=8192), orc.hash_agg(acfg, probe, 4, 4))


def test_operators_run_out_of_the_arena(orc):
    ctx = _lib.Context(0)
    try:
        ctx.reserve(1 << 30)
        assert ctx.arena_stats() == {"size": 1 << 30, "used": 0, "peak": 0}
        _join_and_agg(ctx, orc)
        st = ctx.arena_stats()
        assert st["used"] == 0 and st["peak"] > (1 << 20), st  # the operators' buffers came from the slab and went back
        p = ctx.alloc(1 << 20)
        assert (1 << 20) <= ctx.arena_stats()["used"] <= (1 << 20) + 4096  # (+ the 64 readable bytes behind every tsq_dev_alloc block, rounded to the slab's granule)
        with pytest.raises(_lib.TsqError):  # live buffers: the slab cannot be given back
            ctx.reserve(0)
        ctx.free(p)
        ctx.reserve(0)
        assert ctx.arena_stats()["size"] == 0
        _join_and_agg(ctx, orc)  # and without an arena
    finally:
        ctx.close()


def test_an_arena_that_is_too_small_falls_through(orc):
    ctx = _lib.Context(0)
    try:
        ctx.reserve(1 << 16)  # 64 KB: nothing of substance fits
        _join_and_agg(ctx, orc)
        assert ctx.arena_stats()["used"] == 0
    finally:
        ctx.close()
