#!/usr/bin/env python3
"""bench.py — probed rows/sec on an int64-key inner hash join (BASELINE.json metric).

Workload (N = 1): `SELECT count(*) FROM probe JOIN build ON probe.k = build.k`, 1e8 ⋈ 1e8 rows of
(k int64, v int64) per side, J-uniq-shuffled: build keys are a bijection of
[0, N_b) in pseudo-random order, probe keys are uniform in [0, N_b) (hit ratio 1.0), generated on
the device so the tables never cross PCIe.  The build side is built once and stays resident in
HBM; one "step" = one probe pass of all N_p probe rows through libtsq: radix partition of the probe
keys (k_radix_partition) + partition-at-a-time probe (k_radix_probe_count), or the direct probe
(k_probe_count) with --radix off.
N > 1: weak scaling — every rank owns N_b build and N_p probe rows; rows are redistributed by
hash-radix with an RCCL all-to-all (tinysql_amd/parallel.py); a step = split + exchange + local
probe of the probe side; the build side is redistributed and built once (untimed, resident).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_radix_probe_count):
algorithmic bytes (24 B per probe row: 8 B key + one 16 B slot) / its average
HIP-event duration over the timed steps; `roofline.probe_phase` prices the whole step (partition +
probe) against the same 24 B/row, `roofline.partition` the partition kernel at its own 16 B/key.
`cpu_baseline` = the oracle's C++ restatement of the reference algorithm (oracle/, test
infrastructure — used here only as the reported baseline) timed on the host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--exchange-chunks", type=int, default=4, help="N>1: pieces of the probe-side exchange (probe of piece c overlaps the exchange of c+1..)")
    ap.add_argument("--build-rows", type=int, default=100_000_000)
    ap.add_argument("--probe-rows", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-build-rows", type=int, default=10_000_000)
    ap.add_argument("--cpu-probe-rows", type=int, default=20_000_000)
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path even with one rank (validation)")
    ap.add_argument("--radix", choices=["auto", "off", "force"], default="auto", help="probe strategy (tsq_join_set_radix)")
    args = ap.parse_args()

    n_gpus = args.gpus
    distributed = n_gpus > 1 or args.force_dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch = dist = None
    if distributed:
        # torch FIRST: it bundles its own libamdhip64.so.7; loading it before libtsq makes both share
        # one HIP runtime in this process (see DESIGN.md "Process model").
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
        assert world == n_gpus or args.force_dist, "WORLD_SIZE must equal --gpus"

    from tinysql_amd import _abi as abi
    from tinysql_amd import _lib

    ctx = _lib.Context(local_rank)
    lib = ctx.lib
    if distributed:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    nb, npr = args.build_rows, args.probe_rows  # per GPU
    nb_global = nb * world
    t_setup = time.time()

    def spec(kind, **kw):
        s = abi.GenSpec()
        s.kind, s.seed = kind, 42
        for k, v in kw.items():
            setattr(s, k, v)
        return s

    def dev_col(ptr, n):
        c = abi.Col()
        c.data, c.length, c.elem_size, c.type, c.flags = ptr, n, 8, abi.I64, abi.COL_DEVICE
        return c

    # ---------------------------------------------------------------- tables on device
    if distributed:
        bk_t = torch.empty(nb, dtype=torch.int64, device="cuda")
        bv_t = torch.empty(nb, dtype=torch.int64, device="cuda")
        pk_t = torch.empty(npr, dtype=torch.int64, device="cuda")
        bk, bv, pk = bk_t.data_ptr(), bv_t.data_ptr(), pk_t.data_ptr()
        pv = None
    else:
        bk, bv, pk, pv = (ctx.alloc(nb * 8), ctx.alloc(nb * 8), ctx.alloc(npr * 8), ctx.alloc(npr * 8))
    a_mult = 2654435761  # odd, not a multiple of 5: coprime with 10^k sizes -> bijection on [0, nb_global)
    assert nb_global < (1 << 31)
    ctx.gen_column(spec(abi.GEN_AFFINE, table=2, a=a_mult, b=12345, m=nb_global, start=rank * nb), nb, bk)
    ctx.gen_column(spec(abi.GEN_HASH_OF_COL, table=2, b=0xABCDEF), nb, bv, src=bk)
    ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=nb_global, start=rank * npr), npr, pk)
    if pv:
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 62, start=rank * npr), npr, pv)
    ctx.sync()

    # ---------------------------------------------------------------- build (resident in HBM)
    cfg = abi.JoinCfg()
    cfg.join_type, cfg.build_is_right, cfg.n_keys = abi.JOIN_INNER, 1, 1
    cfg.n_build_cols = 2
    cfg.n_probe_cols = 1 if distributed else 2  # COUNT(*) needs only the key on the exchanged probe side
    for i in range(2):
        cfg.build_types[i] = cfg.probe_types[i] = abi.I64
    cfg.max_chunk_size, cfg.concurrency = 1024, 5
    h = C.c_void_p()
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    _lib.check(lib.tsq_join_set_radix(h, {"auto": abi.RADIX_AUTO, "off": abi.RADIX_OFF, "force": abi.RADIX_FORCE}[args.radix]), h)
    if distributed:
        from tinysql_amd import parallel
        (rbk, rbv), nb_local = parallel.redistribute(ctx, dist, torch, [bk_t, bv_t], [abi.I64, abi.I64], 0, 0, nb)
        torch.cuda.synchronize()
        bcols = (abi.Col * 2)(dev_col(rbk.data_ptr(), nb_local), dev_col(rbv.data_ptr(), nb_local))
    else:
        nb_local = nb
        bcols = (abi.Col * 2)(dev_col(bk, nb), dev_col(bv, nb))
    t0 = time.time()
    _lib.check(lib.tsq_join_build_push(h, bcols, 2, nb_local), h)
    _lib.check(lib.tsq_join_build_finish(h), h)
    ctx.sync()
    build_wall_ms = (time.time() - t0) * 1e3
    _lib.check(lib.tsq_join_set_count_only(h, 1), h)

    if distributed:
        from tinysql_amd import parallel

        def step():
            # the probe of piece c runs while pieces c+1.. are still being exchanged (parallel.redistribute_pipelined)
            # libtsq runs on torch's current stream, so a received tensor may be dropped as soon as its probe is queued:
            # the caching allocator hands the block to later work of the same stream only (retaining every piece until the
            # end of the timed region made each step allocate 1.6 GB of fresh device memory: 17 ms per step).
            for (rpk,), n_local in parallel.redistribute_pipelined(ctx, dist, torch, [pk_t], [abi.I64], 0, 0, npr, args.exchange_chunks):
                if n_local:
                    pc = (abi.Col * 1)(dev_col(rpk.data_ptr(), n_local))
                    _lib.check(lib.tsq_join_probe_push(h, pc, 1, n_local, None), h)
                    local_probe[0] += n_local
                    local_probe[1] += 1

        def full_sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        pcols = (abi.Col * 2)(dev_col(pk, npr), dev_col(pv, npr))

        def step():
            _lib.check(lib.tsq_join_probe_push(h, pcols, 2, npr, None), h)

        def full_sync():
            # single process, single stream: hipStreamSynchronize of the only stream with work
            # (== torch.cuda.synchronize() for this process; torch is not loaded at N=1)
            ctx.sync()

    local_probe = [0, 0]  # N>1: rows this rank probed / probe batches it pushed (for the per-launch roofline figure)
    keep = []
    for _ in range(args.warmup):
        keep.append(step())
    full_sync()
    keep.clear()
    local_probe[0] = local_probe[1] = 0
    setup_s = time.time() - t_setup

    # ---------------------------------------------------------------- timed region: exactly K steps
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep.append(step())
    ev_ms = ctx.timer_stop_ms()  # HIP events on the stream the kernels were launched on
    full_sync()
    elapsed = time.perf_counter() - t0
    keep.clear()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N>1: one extra, untimed pass of the split + all-to-all alone (no probe), for SURVEY.md §8(d)'s t_exchange
    exchange_ms = None
    if distributed:
        try:
            full_sync()
            te = time.perf_counter()
            for _pieces, _n in parallel.redistribute_pipelined(ctx, dist, torch, [pk_t], [abi.I64], 0, 0, npr, args.exchange_chunks):
                pass
            full_sync()
            tx = torch.tensor([time.perf_counter() - te], dtype=torch.float64, device="cuda")
            dist.all_reduce(tx, op=dist.ReduceOp.MAX)
            exchange_ms = float(tx.item()) * 1e3
        except Exception:  # reporting only
            exchange_ms = None

    # ---------------------------------------------------------------- verify (size-independent property)
    cnt = C.c_int64(0)
    _lib.check(lib.tsq_join_count(h, C.byref(cnt)), h)
    total = cnt.value
    if distributed:
        t = torch.tensor([total], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        total = int(t.item())
    # build keys are a bijection of [0, nb_global), every probe key lies in [0, nb_global): each probe row joins once
    expect = (args.steps + args.warmup) * npr * world
    ok = total == expect
    st = abi.Stats()
    _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
    lib.tsq_join_destroy(h)

    rows_per_s = npr * world * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    step_ev_ms = ev_ms / args.steps  # HIP events around the K steps on the launch stream
    algo_bytes = 24.0 * npr
    radix = st.radix_batches > 0
    if radix and st.radix_timed_batches > 0:
        nt = min(st.radix_timed_batches, args.steps)  # the event ring keeps the most recent batches = the timed steps
        kernel_name = "k_radix_probe_count<2,0>"
        kernel_ms = st.radix_probe_kernel_ms_sum / st.radix_timed_batches
        part_ms = st.partition_kernel_ms_sum / st.radix_timed_batches
    else:
        nt = args.steps
        kernel_name = "k_probe_count<MULTI=0,GEN=0,CHK=0>"
        kernel_ms, part_ms = step_ev_ms, 0.0
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9

    out = {
        "metric": "probed rows/sec on int64-key inner hash join",
        "value": rows_per_s,
        "unit": "rows/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": "SELECT count(*) FROM probe JOIN build ON k: %.0e x %.0e int64-key inner hash join per GPU, "
                        "J-uniq-shuffled, hit ratio 1.0, build side resident in HBM" % (npr, nb),
            "probe_rows_per_gpu": npr, "build_rows_per_gpu": nb,
            "parallelism": "hash-radix x%d, RCCL all-to-all" % world if distributed else "single GPU",
        },
        "verified": bool(ok),
        "joined_rows": total,
        "build_ms": build_wall_ms,
        "build_kernel_ms": st.build_kernel_ms,
        "build_strategy": "partitioned: 2 radix passes + LDS slice images" if st.build_partitioned else "row-at-a-time CAS",
        "table_bytes": st.table_bytes,
        "setup_s": setup_s,
    }
    if exchange_ms is not None:
        out["split_and_exchange_ms"] = exchange_ms  # tsq_radix_split + RCCL all-to-all of one step's probe keys, without the probes
    out["probe_strategy"] = ("radix 2^%d partitions" % st.radix_bits) if radix else "direct"
    traffic = traffic_part = None
    try:  # PMC-derived HBM bytes per launch are measured offline (rocprofv3 --pmc passes) and committed under profiles/
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
        w = tj["workload"]
        if w["probe_rows"] == npr and w["build_rows"] == nb and world == 1:
            if radix and w["radix_bits"] == st.radix_bits:
                traffic = tj["k_radix_probe_count<2,0>"]["traffic_bytes"]
                traffic_part = tj["k_radix_partition<1024,16,4,0,false>"]["traffic_bytes"]
            elif not radix:
                traffic = tj["k_probe_count<false,false,false> (direct probe, --radix off)"]["traffic_bytes"]
    except Exception:
        pass
    if not distributed:
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_source": "profiles/traffic_r01.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)" if traffic else None,
            "kernel": kernel_name, "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": algo_bytes,
            "probe_phase": {"ms": step_ev_ms, "achieved": algo_bytes / (step_ev_ms * 1e-3) / 1e9,
                            "frac": algo_bytes / (step_ev_ms * 1e-3) / 1e9 / 8000.0,
                            "note": "whole step (memsets + partition + probe + overflow kernels) priced at 24 B/probe row"},
        }
        if radix:
            pb = 16.0 * npr  # 8 B key read + 8 B key written per probe row (COUNT(*) carries no payload)
            out["roofline"]["partition"] = {"kernel": "k_radix_partition<1024,16,4,0,false>", "kernel_ms": part_ms,
                                            "algorithmic_bytes_per_launch": pb, "achieved": pb / (part_ms * 1e-3) / 1e9,
                                            "frac": pb / (part_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic_part}
            out["radix_overflow_rows"] = st.radix_overflow_rows
    elif radix and local_probe[1] > 0 and kernel_ms > 0:
        # N>1: the local probe runs once per received piece; price the same kernel per launch on rank 0's own pieces
        rows_per_launch = local_probe[0] / local_probe[1]
        ab = 24.0 * rows_per_launch
        out["roofline"] = {
            "bound": "hbm", "achieved": ab / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": ab / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
            "kernel": kernel_name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": ab,
            "note": "rank 0, per probe launch (one launch per received piece of %.3g rows on average, %d pieces per step); "
                    "the step also contains tsq_radix_split and the RCCL all-to-all, which this figure does not price"
                    % (rows_per_launch, args.exchange_chunks),
            "partition": {"kernel": "k_radix_partition<1024,16,4,0,false>", "kernel_ms": part_ms,
                          "algorithmic_bytes_per_launch": 16.0 * rows_per_launch,
                          "achieved": 16.0 * rows_per_launch / (part_ms * 1e-3) / 1e9 if part_ms > 0 else None},
        }

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(abi, args.cpu_build_rows, args.cpu_probe_rows)
        except Exception as e:  # the baseline is reporting only; never fail the bench for it
            out["cpu_baseline"] = {"error": str(e)[:200]}

    if not distributed:
        for p in (bk, bv, pk, pv):
            ctx.free(p)
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if not ok:
        sys.exit("bench: join count mismatch: got %d expected %d" % (total, expect))


def cpu_baseline(abi, nb, npr):
    """oracle = C++ restatement of the reference Go algorithm (Go toolchain unavailable; the reference
    operator bodies are course stubs): single-threaded build, `threads` probe workers, 1024-row chunks."""
    import numpy as np

    from oracle import binding as orc
    from tinysql_amd.chunk import Chunk, Column

    def spec(kind, **kw):
        s = abi.GenSpec()
        s.kind, s.seed = kind, 42
        for k, v in kw.items():
            setattr(s, k, v)
        return s

    bk, _ = orc.gen_column(spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=nb), nb)
    bv, _ = orc.gen_column(spec(abi.GEN_HASH_OF_COL, table=2, b=0xABCDEF), nb, src=bk)
    pk, _ = orc.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=nb), npr)
    pv, _ = orc.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 62), npr)
    build = Chunk([Column(abi.I64, bk.view(np.int64)), Column(abi.I64, bv.view(np.int64))])
    probe = Chunk([Column(abi.I64, pk.view(np.int64)), Column(abi.I64, pv.view(np.int64))])
    cfg = abi.JoinCfg()
    cfg.join_type, cfg.build_is_right, cfg.n_keys = abi.JOIN_INNER, 1, 1
    cfg.n_build_cols = cfg.n_probe_cols = 2
    for i in range(2):
        cfg.build_types[i] = cfg.probe_types[i] = abi.I64
    cfg.max_chunk_size = 1024
    cfg.est_build_rows = nb
    cores = os.cpu_count() or 1
    threads = min(cores, 5)  # tidb_hash_join_concurrency default = 5 (sessionctx/variable/tidb_vars.go:249)
    n, build_ms, probe_ms, _, _ = orc.hash_join_timed(cfg, build, probe, threads)
    assert n == npr
    return {
        "value": npr / (probe_ms * 1e-3), "unit": "rows/s", "cores": threads, "kind": "port",
        "sample": "C++ restatement of the reference HashJoinExec algorithm (FNV-1 + chained map, 1024-row chunks), "
                  "%d probe worker threads of %d host cores, %.0e probe rows x %.0e build rows of the same generators; "
                  "build %.0f ms single-threaded, probe %.0f ms" % (threads, cores, npr, nb, build_ms, probe_ms),
        "build_rows_per_s": nb / (build_ms * 1e-3),
    }


if __name__ == "__main__":
    main()
