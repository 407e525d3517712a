"""One rank of the multi-GPU parity run (tests/test_comm_gpu.py): the REAL path — tsq_radix_split + RCCL send/recv inside
libtsq (tsq_redistribute) + the HIP join / aggregate — against the oracle's whole-table result.
Every rank generates ALL ranks' rows from per-rank seeds (so it can run the oracle on the union), feeds its own rows to the
distributed plan, and checks: the all-reduced join count equals the oracle's; every received key ranks to this rank; the
final groups this rank owns equal the oracle's groups with those keys; the group counts of all ranks add up.
usage: dist_gpu_worker.py  (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT in the environment)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402
from tests import helpers as H  # noqa: E402
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402
from tinysql_amd import parallel  # noqa: E402
from tinysql_amd.chunk import Chunk, Column  # noqa: E402


def np_rank(keys, parts):
    k = keys.view(np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xFF51AFD7ED558CCD)
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xC4CEB9FE1A85EC53)
        k ^= k >> np.uint64(33)
    return (((k & np.uint64(0xFFFF)) * np.uint64(parts)) >> np.uint64(16)).astype(np.int64)


def rows_of(rank, nb, npr):
    rng = np.random.default_rng(7000 + rank)
    return (rng.integers(0, 60_000, nb).astype(np.int64), rng.integers(-99, 99, nb).astype(np.int64),
            rng.integers(0, 70_000, npr).astype(np.int64), rng.integers(-99, 99, npr).astype(np.int64))


def dev(ctx, arr, keep):
    p = ctx.alloc(max(arr.nbytes, 8))
    ctx.h2d(p, np.ascontiguousarray(arr))
    keep.append(p)
    c = abi.Col()
    c.data, c.length, c.elem_size, c.type, c.flags = p, len(arr), 8, abi.I64, abi.COL_DEVICE
    return c


def devn(ctx, arr, notnull, keep):
    """device column with a null bitmap (bit = 1: NOT NULL)"""
    c = dev(ctx, arr, keep)
    bm = np.concatenate([np.packbits(notnull.astype(np.uint8), bitorder="little"), np.zeros(8, np.uint8)])
    q = ctx.alloc(bm.nbytes)
    ctx.h2d(q, bm)
    keep.append(q)
    c.null_bitmap = q
    return c


def null_rows_of(rank, n):
    rng = np.random.default_rng(9100 + rank)
    k = rng.integers(0, 5000, n).astype(np.int64)
    v = rng.integers(-99, 99, n).astype(np.int64)
    kn = rng.random(n) > 0.03
    vn = rng.random(n) > 0.10
    vn[k % 17 == 0] = False  # whole groups without a single non-NULL value: SUM / MIN are NULL, also as partial results
    return k, kn, v, vn


def main():
    import time
    t_start = time.time()

    def lap(what):
        if os.environ.get("TSQ_TEST_TIMING"):
            print("[rank %s] %-28s %.2f s" % (os.environ.get("RANK", "0"), what, time.time() - t_start), flush=True)

    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    nb, npr = 150_000, 400_000  # per rank: above the fast-split threshold (2^16 rows), ragged pieces
    allrows = [rows_of(r, nb, npr + 1000 * r) for r in range(world)]
    bk, bv, pk, pv = allrows[rank]
    keep = []
    with _lib.Context(local) as ctx:
        lap("context")
        comm = parallel.Comm(ctx, rank, world)
        lap("communicator")
        try:
            assert comm.allreduce_i64([rank + 1, 10])[0] == world * (world + 1) // 2
            assert comm.allreduce_i64([rank], parallel.Comm.MAX)[0] == world - 1
            assert abs(comm.allreduce_f64([0.5 * (rank + 1)])[0] - 0.25 * world * (world + 1)) < 1e-12
            # ---- redistribute alone: every received key ranks here, nothing lost
            got, n = comm.redistribute([dev(ctx, bk, keep), dev(ctx, bv, keep)], 0, 0, nb, slot=3)
            comm.wait(3)
            ctx.sync()
            rk, rv = np.empty(n, np.int64), np.empty(n, np.int64)
            if n:
                ctx.d2h(rk, got[0].data)
                ctx.d2h(rv, got[1].data)
            assert (np_rank(rk, world) == rank).all()
            uk = np.concatenate([a[0] for a in allrows])
            uv = np.concatenate([a[1] for a in allrows])
            own = np_rank(uk, world) == rank
            assert sorted(zip(rk.tolist(), rv.tolist())) == sorted(zip(uk[own].tolist(), uv[own].tolist()))
            assert comm.allreduce_i64([n])[0] == nb * world
            lap("redistribute checked")
            # ---- distributed COUNT(*) join vs the oracle on the union of all ranks' rows
            t = [abi.I64, abi.I64]
            cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_INNER, 1)
            j = parallel.DistHashJoinCount(comm, cfg)
            try:
                j.build([dev(ctx, bk, keep), dev(ctx, bv, keep)], 0, nb)
                j.probe([dev(ctx, pk, keep), dev(ctx, pv, keep)], 0, len(pk), n_pieces=3)
                j.probe([dev(ctx, pk, keep), dev(ctx, pv, keep)], 0, len(pk), n_pieces=1)
                j.probe([dev(ctx, pk, keep), dev(ctx, pv, keep)], 0, len(pk), n_pieces=3, batched_counts=True)  # one count exchange for the pieces
                total = j.count()
            finally:
                j.close()
            lap("distributed join")
            ubk = np.concatenate([a[0] for a in allrows])
            ubv = np.concatenate([a[1] for a in allrows])
            upk = np.concatenate([a[2] for a in allrows])
            upv = np.concatenate([a[3] for a in allrows])
            want = orc.hash_join(cfg, Chunk([Column(abi.I64, ubk), Column(abi.I64, ubv)]), Chunk([Column(abi.I64, upk), Column(abi.I64, upv)])).NumRows()
            assert total == 3 * want, (total, want)
            lap("oracle join")
            # ---- the SHARED-IMAGES plan (tsq_join_build_finish_shared): the packed images of the whole build side are summed across the
            # ranks once, every rank probes its OWN rows — no probe row is exchanged.  Route asserted, count vs the oracle's whole join.
            def shared_case(bkeys_of, pkeys_of, expect_shared, what, expect_bits=None):
                allb = [bkeys_of(r) for r in range(world)]
                allp = [pkeys_of(r) for r in range(world)]
                js = parallel.DistHashJoinCount(comm, cfg, packing_mode=abi.RADIX_FORCE)
                try:
                    mb, mp = allb[rank], allp[rank]
                    js.build([dev(ctx, mb, keep), dev(ctx, mb * 2, keep)], 0, len(mb))
                    assert js.shared == expect_shared, (what, js.shared)
                    js.probe([dev(ctx, mp, keep), dev(ctx, mp + 1, keep)], 0, len(mp), n_pieces=2)
                    js.probe([dev(ctx, mp, keep), dev(ctx, mp + 1, keep)], 0, len(mp), n_pieces=1)
                    tot = js.count()
                    stt = js.stats()
                finally:
                    js.close()
                ub, up = np.concatenate(allb), np.concatenate(allp)
                want_s = orc.hash_join(cfg, Chunk([Column(abi.I64, ub), Column(abi.I64, ub * 2)]), Chunk([Column(abi.I64, up), Column(abi.I64, up + 1)])).NumRows()
                assert tot == 2 * want_s, (what, tot, want_s)
                if expect_shared:
                    assert stt.shared_build == 1 and stt.probe_route == abi.ROUTE_PACKED and stt.shared_image_bytes > 0, (what, stt.shared_build, stt.probe_route)
                    assert js.wire_bytes_probe == 0
                    if expect_bits is not None:
                        assert (stt.packed_key_bits > 28) == expect_bits, (what, stt.packed_key_bits)
                else:
                    assert stt.shared_build == 0
                return want_s
            g = lambda seed: np.random.default_rng(seed)  # noqa: E731
            # byte cells: duplicate keys inside a rank and across ranks (cells add up), negative keys, probe keys outside the range
            w1 = shared_case(lambda r: g(8100 + r).integers(-30_000, 40_000, 120_000 + 999 * r).astype(np.int64),
                             lambda r: g(8200 + r).integers(-50_000, 60_000, 300_000 + 77 * r).astype(np.int64), True, "byte cells", expect_bits=False)
            assert w1 > 0
            # bit cells: a UNIQUE build side over the ranks whose keys span 30 bits (rank r owns the keys = r mod world)
            uniq = lambda r: (g(8300).permutation(1 << 20)[:200_000].astype(np.int64) * world + r) * (1 << 9) + 5  # noqa: E731
            w2 = shared_case(uniq, lambda r: np.concatenate([uniq((r + 1) % world)[:150_000], g(8400 + r).integers(0, 1 << 30, 100_000).astype(np.int64)]), True,
                             "bit cells", expect_bits=True)
            assert w2 >= 150_000 * world
            # a key with more than 255 build rows over the ranks TOGETHER (each rank alone stays below): the summed byte wraps, every rank
            # notices in the population of the sum and all of them take the exchange plan
            hot = 300 // world + 1
            shared_case(lambda r: np.concatenate([np.full(hot, 777, np.int64), g(8500 + r).integers(0, 50_000, 100_000).astype(np.int64)]),
                        lambda r: g(8600 + r).integers(0, 50_000, 200_000).astype(np.int64), False, "a cell beyond 255")
            if world > 1:
                # a key present on two ranks of a wide (bit-cell) range: a carry in the summed words -> the exchange plan, on every rank
                shared_case(lambda r: uniq(r) if r == 0 else np.concatenate([uniq(r), uniq(0)[:1]]),
                            lambda r: uniq(r)[:1000], False, "a key on two ranks, bit cells")
                # a rank without build rows takes part in every collective
                shared_case(lambda r: g(8700 + r).integers(0, 40_000, 0 if r == 0 else 90_000).astype(np.int64),
                            lambda r: g(8800 + r).integers(0, 40_000, 100_000).astype(np.int64), True, "a rank without build rows")
            lap("shared images")
            # ---- distributed GROUP BY k: SUM(v), COUNT(*), MIN(v) vs the oracle
            paggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_SUM, 1, abi.I64, abi.MODE_PARTIAL1),
                     (abi.AGG_COUNT, -1, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_MIN, 1, abi.I64, abi.MODE_PARTIAL1)]
            ptypes = [abi.I64, abi.I64, abi.I64, abi.I64]
            faggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_FINAL), (abi.AGG_SUM, 1, abi.I64, abi.MODE_FINAL),
                     (abi.AGG_COUNT, 2, abi.I64, abi.MODE_FINAL), (abi.AGG_MIN, 3, abi.I64, abi.MODE_FINAL)]
            out, ng = parallel.dist_hash_agg(comm, H.agg_cfg(t, [0], paggs), H.agg_cfg(ptypes, [0], faggs), [dev(ctx, pk, keep), dev(ctx, pv, keep)],
                                             len(pk), ptypes)
            lap("distributed aggregate")
            cols = []
            for p, q in out:
                a = np.empty(ng, np.int64)
                if ng:
                    ctx.d2h(a, p)
                cols.append(a)
                ctx.free(p)
                ctx.free(q)
            caggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_MIN, 1, abi.I64)]
            whole = orc.hash_agg(H.agg_cfg(t, [0], caggs), Chunk([Column(abi.I64, upk), Column(abi.I64, upv)]), 4, 4)
            wk = np.array(whole.columns[0].data)
            mine = np_rank(wk, world) == rank
            want_rows = sorted(zip(*[np.array(c.data)[mine].tolist() for c in whole.columns]))
            assert sorted(zip(*[c.tolist() for c in cols])) == want_rows
            assert comm.allreduce_i64([ng])[0] == whole.NumRows()
            lap("oracle aggregate")
            # ---- NULLs travel with the rows: NULL keys form one group (owned by rank 0), NULL values and NULL partial results
            nn_rows = [null_rows_of(r, 90_000 + 500 * r) for r in range(world)]
            k, kn, v, vn = nn_rows[rank]
            got, n = comm.redistribute([devn(ctx, k, kn, keep), devn(ctx, v, vn, keep)], 0, 1, len(k), slot=5)
            comm.wait(5)
            ctx.sync()
            rk, rv = np.empty(n, np.int64), np.empty(n, np.int64)
            bk2, bv2 = np.zeros((n + 7) // 8 + 1, np.uint8), np.zeros((n + 7) // 8 + 1, np.uint8)
            if n:
                ctx.d2h(rk, got[0].data)
                ctx.d2h(rv, got[1].data)
                ctx.d2h(bk2[:(n + 7) // 8], got[0].null_bitmap)
                ctx.d2h(bv2[:(n + 7) // 8], got[1].null_bitmap)
            rkn = np.unpackbits(bk2, bitorder="little")[:n].astype(bool)
            rvn = np.unpackbits(bv2, bitorder="little")[:n].astype(bool)
            uk2 = np.concatenate([a[0] for a in nn_rows])
            ukn = np.concatenate([a[1] for a in nn_rows])
            uv2 = np.concatenate([a[2] for a in nn_rows])
            uvn = np.concatenate([a[3] for a in nn_rows])
            owner = np.where(ukn, np_rank(uk2, world), 0)
            cell = lambda x, ok: [a if b else None for a, b in zip(x.tolist(), ok.tolist())]  # noqa: E731
            key_of = lambda r: tuple((x is None, x or 0) for x in r)  # noqa: E731
            mine = owner == rank
            assert sorted(zip(cell(rk, rkn), cell(rv, rvn)), key=key_of) == sorted(zip(cell(uk2[mine], ukn[mine]), cell(uv2[mine], uvn[mine])), key=key_of)
            paggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_SUM, 1, abi.I64, abi.MODE_PARTIAL1),
                     (abi.AGG_COUNT, 1, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_MIN, 1, abi.I64, abi.MODE_PARTIAL1)]
            faggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_FINAL), (abi.AGG_SUM, 1, abi.I64, abi.MODE_FINAL),
                     (abi.AGG_COUNT, 2, abi.I64, abi.MODE_FINAL), (abi.AGG_MIN, 3, abi.I64, abi.MODE_FINAL)]
            out, ng2 = parallel.dist_hash_agg(comm, H.agg_cfg(t, [0], paggs), H.agg_cfg(ptypes, [0], faggs),
                                              [devn(ctx, k, kn, keep), devn(ctx, v, vn, keep)], len(k), ptypes)
            got_cols = []
            for p, q in out:
                a, bm = np.empty(ng2, np.int64), np.zeros((ng2 + 7) // 8 + 1, np.uint8)
                if ng2:
                    ctx.d2h(a, p)
                    ctx.d2h(bm[:(ng2 + 7) // 8], q)
                got_cols.append(cell(a, np.unpackbits(bm, bitorder="little")[:ng2].astype(bool)))
                ctx.free(p)
                ctx.free(q)
            caggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64)]
            whole2 = orc.hash_agg(H.agg_cfg(t, [0], caggs), Chunk([Column(abi.I64, uk2, ukn), Column(abi.I64, uv2, uvn)]), 4, 4)
            wrows = whole2.rows()
            wown = [0 if r[0] is None else int(np_rank(np.array([r[0]], np.int64), world)[0]) for r in wrows]
            assert sorted(zip(*got_cols), key=key_of) == sorted([r for r, o in zip(wrows, wown) if o == rank], key=key_of)
            assert comm.allreduce_i64([ng2])[0] == whole2.NumRows()
            assert any(r[1] is None for r in wrows) and any(r[0] is None for r in wrows)  # the case really holds NULL sums and the NULL group
            lap("NULLs through the exchange")
            # ---- var-len payload columns travel with their rows (offsets slices + bytes per run, rebased on arrival)
            from tinysql_amd.chunk import StrColumn
            def str_rows_of(r, n):
                g = np.random.default_rng(9900 + r)
                keys = g.integers(0, 1 << 40, n).astype(np.int64)
                names = [None if g.random() < 0.1 else (b"" if g.random() < 0.1 else b"r%d-%d-" % (r, i) + bytes([97 + i % 26]) * int(g.integers(0, 40))) for i in range(n)]
                notes = [bytes(g.integers(0, 256, int(g.integers(0, 9)), dtype=np.uint8)) for _ in range(n)]
                return keys, names, notes
            srows = [str_rows_of(r, 30_000 + 700 * r) for r in range(world)]
            sk, snames, snotes = srows[rank]
            cn, cm = G.DevStrCol(ctx, StrColumn(snames)), G.DevStrCol(ctx, StrColumn(snotes))
            got, n = comm.redistribute([dev(ctx, sk, keep), cn.col(), cm.col(), dev(ctx, sk * 3, keep)], 0, 0, len(sk), slot=6)
            comm.wait(6)
            ctx.sync()
            def pull_str(col, n):
                offs = np.zeros(n + 1, np.int64)
                ctx.d2h(offs, col.offsets)
                data = np.zeros(int(offs[n]) + 8, np.uint8)
                if offs[n]:
                    ctx.d2h(data[:int(offs[n])], col.data)
                nn = np.ones(n, bool)
                if col.null_bitmap:
                    bm = np.zeros((n + 7) // 8 + 1, np.uint8)
                    ctx.d2h(bm[:(n + 7) // 8], col.null_bitmap)
                    nn = np.unpackbits(bm, bitorder="little")[:n].astype(bool)
                raw = data.tobytes()
                assert offs[0] == 0 and (np.diff(offs) >= 0).all()
                return [raw[int(offs[i]):int(offs[i + 1])] if nn[i] else None for i in range(n)]
            rk, r3 = np.empty(n, np.int64), np.empty(n, np.int64)
            if n:
                ctx.d2h(rk, got[0].data)
                ctx.d2h(r3, got[3].data)
            gn, gm = pull_str(got[1], n), pull_str(got[2], n)
            want = []
            for r in range(world):
                k, a, b = srows[r]
                mine = np_rank(k, world) == rank
                want += [(int(k[i]), a[i], b[i], int(k[i]) * 3) for i in np.nonzero(mine)[0]]
            skey = lambda t: (t[0], t[1] is None, t[1] or b"", t[2])  # noqa: E731
            assert n == len(want) and sorted(zip(rk.tolist(), gn, gm, r3.tolist()), key=skey) == sorted(want, key=skey)
            # a string KEY: rows with equal names meet on one rank (whichever), every row arrives exactly once
            got, n = comm.redistribute([cn.col(), dev(ctx, sk, keep)], 0, 1, len(sk), slot=7)
            comm.wait(7)
            ctx.sync()
            gn2 = pull_str(got[0], n)
            rk2 = np.empty(n, np.int64)
            if n:
                ctx.d2h(rk2, got[1].data)
            mine_names = set(gn2)
            counts = comm.allreduce_i64([n, len(mine_names)])
            all_names = set()
            for r in range(world):
                all_names |= set(srows[r][1])
            assert counts[0] == sum(len(srows[r][0]) for r in range(world)) and counts[1] == len(all_names)  # no name on two ranks
            union = {}
            for r in range(world):
                for k, a in zip(srows[r][0].tolist(), srows[r][1]):
                    union.setdefault(a, []).append(k)
            for name in list(mine_names)[:200]:
                assert sorted(k for k, a in zip(rk2.tolist(), gn2) if a == name) == sorted(union[name])
            # ---- GROUP BY a string across the ranks: partial rows (name, sum, count) shuffled by rank(name), merged by their owner
            def grows_of(r):
                n_r = len(srows[r][0])
                return [None if (i + r) % 41 == 0 else b"g%03d" % ((i * 7 + r) % 613) for i in range(n_r)], (srows[r][0] % 1000).astype(np.int64)
            gnames, gv = grows_of(rank)
            cg = G.DevStrCol(ctx, StrColumn(gnames))
            st = [abi.BYTES, abi.I64]
            spt = [abi.BYTES, abi.I64, abi.I64]
            spaggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES, abi.MODE_PARTIAL1), (abi.AGG_SUM, 1, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_COUNT, 1, abi.I64, abi.MODE_PARTIAL1)]
            sfaggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES, abi.MODE_FINAL), (abi.AGG_SUM, 1, abi.I64, abi.MODE_FINAL), (abi.AGG_COUNT, 2, abi.I64, abi.MODE_FINAL)]
            sout, sng = parallel.dist_hash_agg(comm, H.agg_cfg(st, [0], spaggs), H.agg_cfg(spt, [0], sfaggs), [cg.col(), dev(ctx, gv, keep)], len(gv), spt,
                                               out_types=[abi.BYTES, abi.I64, abi.I64])
            class _V:  # a view of a returned var-len column for pull_str
                pass
            v0 = _V()
            v0.data, v0.null_bitmap, v0.offsets = sout[0][0], sout[0][1], sout[0][2]
            names_out = pull_str(v0, sng)
            sums, cnts = np.empty(sng, np.int64), np.empty(sng, np.int64)
            if sng:
                ctx.d2h(sums, sout[1][0])
                ctx.d2h(cnts, sout[2][0])
            wantg = {}
            for r in range(world):
                a, b = grows_of(r)
                for nm, x in zip(a, b.tolist()):
                    e = wantg.setdefault(nm, [0, 0])
                    e[0] += x
                    e[1] += 1
            mine_g = dict(zip(names_out, zip(sums.tolist(), cnts.tolist())))
            assert len(mine_g) == sng and all(tuple(wantg[k]) == v for k, v in mine_g.items())
            assert comm.allreduce_i64([sng])[0] == len(wantg) and (None in wantg)
            for b in sout:
                for q in b[:3]:
                    ctx.free(q)
            cg.free()
            cn.free()
            cm.free()
            lap("strings through the exchange")
            print("rank %d/%d OK: join %d rows, %d of %d groups, %d of %d groups with NULLs" % (rank, world, total, ng, whole.NumRows(), ng2, whole2.NumRows()),
                  flush=True)
        finally:
            for p in keep:
                ctx.free(p)
            comm.close()


if __name__ == "__main__":
    main()
