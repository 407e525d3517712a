#!/usr/bin/env python3
"""bench.py — probed rows/sec on an int64-key inner hash join (BASELINE.json metric).

Workload (N = 1): `SELECT count(*) FROM probe JOIN build ON probe.k = build.k`, 1e8 x 1e8 rows of
(k int64, v int64) per side, J-uniq-shuffled: build keys are a bijection of
[0, N_b) in pseudo-random order, probe keys are uniform in [0, N_b) (hit ratio 1.0), generated on
the device so the tables never cross PCIe.  The build side is built once and stays resident in
HBM; one "step" = one probe pass of all N_p probe rows through libtsq.  On this shape the packed-key
route runs (csrc/tsq_dajoin.h): k_da_partition2 (8-byte key read, 2-byte entry written per probe
row) + k_da_probe_count (entries against one-byte direct-address images in LDS).
N > 1: weak scaling — every rank owns N_b build and N_p probe rows.  Plan "shared images"
(tsq_join_build_finish_shared): the packed images of the whole build side are summed across the ranks
once per build, a step = every rank probing its OWN probe rows, nothing on xGMI; build sides the
images cannot hold fall back to the hash-radix exchange of both sides (tsq_redistribute).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step: `achieved` =
its ALGORITHMIC bytes per launch / its average HIP-event duration over the timed steps, `traffic` =
its HBM bytes per launch from the PMC counters (profiles/traffic_r04.json), `traffic_frac` =
traffic / duration / 8 TB/s.  `roofline.step` prices the whole step the same two ways, plus — labelled —
SURVEY.md 8(d)'s 24 B per probe row, which no longer describes the bytes this algorithm moves.
`cpu_baseline` = the oracle's C++ restatement of the reference algorithm (oracle/, test
infrastructure — used here only as the reported baseline) timed on the host cores on a bounded sample,
at the reference's worker counts (4, 5) and with every core.
N = 1 also reports extra keys measured after the timed region, each verified by a closed form or by
numpy: BASELINE configs[1] and [2], the 8(d) variants (duplicate build keys, hit ratio 0.1, NULL probe
keys), the other routes (64-bit words, bit cells, several key columns) and the materialising join.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def arm_watchdog(seconds, emit):
    """After `seconds` without disarm(): emit() (rank 0 prints what it has), then the process leaves at once — a hung collective cannot be
    cancelled from the thread that sits in it.  Returns disarm()."""
    import threading
    done = threading.Event()

    def run():
        if not done.wait(seconds):
            try:
                emit()
            finally:
                os._exit(0)
    threading.Thread(target=run, daemon=True).start()
    return done.set


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--exchange-chunks", type=int, default=4, help="N>1: pieces of the probe-side exchange (probe of piece c overlaps the exchange of c+1..)")
    ap.add_argument("--build-rows", type=int, default=100_000_000)
    ap.add_argument("--probe-rows", type=int, default=100_000_000)
    ap.add_argument("--rows-global", type=int, default=0, help="TOTAL rows per side over all ranks (strong scaling; BASELINE configs[3] = "
                    "--gpus 8 --rows-global 1000000000); default: --build-rows / --probe-rows PER rank (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-build-rows", type=int, default=20_000_000)
    ap.add_argument("--cpu-probe-rows", type=int, default=30_000_000)
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path even with one rank (validation)")
    ap.add_argument("--emulate-world", type=int, default=0, help="with --force-dist on ONE GPU: generate rank 0's share of a W-rank weak-scaling job "
                    "(keys drawn from the W x build-rows global domain), so the per-rank probe step of an N = W run is measured without W GPUs")
    ap.add_argument("--dist-plan", choices=["auto", "exchange"], default="auto", help="N>1: auto = shared packed images when the build side is packable "
                    "(no probe row crosses xGMI), else the hash-radix exchange; exchange = always redistribute both sides by rank(key)")
    ap.add_argument("--radix", choices=["auto", "off", "force"], default="auto", help="probe strategy (tsq_join_set_radix)")
    ap.add_argument("--packing", choices=["auto", "off"], default="auto", help="key packing of the radix probe (tsq_join_set_key_packing)")
    ap.add_argument("--arena-gb", type=float, default=64.0, help="tsq_ctx_reserve: HBM reserved for the context's operators at start-up, before any "
                    "query runs (0: none — every first allocation of a size is a hipMalloc, ~35 ms per GB)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE", help="tsq_ctx_set_knob before anything runs (A/B measurements), e.g. DA_PARTITION=2")
    ap.add_argument("--only-extras", default="", help="comma-separated keys of the side measurements to run (default: all)")
    ap.add_argument("--no-extras", action="store_true", help="skip the c2 / c3 / materialising side measurements (N = 1)")
    ap.add_argument("--extras-file", default="bench_extras.json", help="name (under gpurun_out/) of the file the whole record is written to")
    args = ap.parse_args()

    n_gpus = args.gpus
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher above us: be the launcher
        self_launch(args)
    distributed = n_gpus > 1 or args.force_dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == n_gpus or args.force_dist or not distributed, "WORLD_SIZE must equal --gpus"
    if os.environ.get("TSQ_BENCH_ECHO_RANK"):  # tests/test_bench_line_cpu.py: the self-launcher's ranks, seen without a GPU
        sys.stderr.write("bench-rank %d of %d local %d\n" % (rank, world, local_rank))
        sys.exit(0)

    from tinysql_amd import _abi as abi
    from tinysql_amd import _lib

    # The contract: rank 0 prints ONE JSON line on stdout.  Libraries below us may print there too (RCCL announces its version on stdout
    # when a communicator is created): until the line is ready, file descriptor 1 is stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # One process per GPU.  N > 1: the ranks meet through libtsq's own communicator (RCCL inside the library, tsq_comm_*):
    # no torch in this process, every call in the timed region is a C-ABI call a Go host could make.
    ctx = _lib.Context(local_rank)
    lib = ctx.lib
    for kv in args.knob:
        name, val = kv.split("=")
        ctx.set_knob(getattr(abi, "KNOB_" + name), int(val))
    t_arena = time.time()
    if args.arena_gb > 0:  # what a host process does once, when it creates its context (untimed set-up, like the table generation)
        try:
            ctx.reserve(int(args.arena_gb * (1 << 30)))
        except Exception as e:  # (a part with less HBM: run without the arena — pool + hipMalloc, as before)
            sys.stderr.write("bench: tsq_ctx_reserve(%g GB) failed (%s): running without an arena\n" % (args.arena_gb, str(e)[:120]))
            args.arena_gb = 0.0
    t_arena = time.time() - t_arena
    comm = None
    if distributed:
        from tinysql_amd import parallel
        comm = parallel.Comm(ctx, rank, world)

    nb, npr = args.build_rows, args.probe_rows  # per GPU
    if args.rows_global > 0:  # the whole join is fixed, every rank starts with 1 / world of both sides (rows are multiples of 64)
        nb = npr = max(64, (args.rows_global // world) & ~63)
    nb_global = nb * world
    emu = args.emulate_world if (args.emulate_world > 1 and world == 1 and distributed) else 0
    if emu:  # rank 0 of an emu-rank job: its 1 / emu of the build side (keys all over the GLOBAL domain), its own probe rows
        nb_global = nb * emu
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
    bk, bv, pk = ctx.alloc(nb * 8), ctx.alloc(nb * 8), ctx.alloc(npr * 8)
    pv = None if distributed else ctx.alloc(npr * 8)
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
    radix_mode = {"auto": abi.RADIX_AUTO, "off": abi.RADIX_OFF, "force": abi.RADIX_FORCE}[args.radix]
    dj = None
    if distributed:
        dj = parallel.DistHashJoinCount(comm, cfg, radix_mode=radix_mode, packing_mode=abi.RADIX_OFF if args.packing == "off" else None,
                                        shared=args.dist_plan == "auto" and args.packing != "off")
        t0 = time.time()
        # shared images: local rows pushed, the packed images summed across the ranks (ONE all-reduce per build side);
        # exchange plan: redistribute by rank(key), then the local build
        nb_local = dj.build([dev_col(bk, nb), dev_col(bv, nb)], 0, nb)
        ctx.sync()
        build_wall_ms = (time.time() - t0) * 1e3
        h = dj.h
        pcols1 = [dev_col(pk, npr)]

        def step():
            # piece c + 1 is split and on the wire (RCCL, the communicator's stream) while piece c is probed (tsq_redistribute /
            # tsq_redistribute_wait / tsq_join_probe_push, tinysql_amd/parallel.py: redistribute_pieces)
            dj.probe(pcols1, 0, npr, args.exchange_chunks)

        def full_sync():
            ctx.sync()
            comm.barrier()
            ctx.sync()
    else:
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        _lib.check(lib.tsq_join_set_radix(h, radix_mode), h)
        if args.packing == "off":
            _lib.check(lib.tsq_join_set_key_packing(h, abi.RADIX_OFF), h)
        nb_local = nb
        bcols = (abi.Col * 2)(dev_col(bk, nb), dev_col(bv, nb))
        t0 = time.time()
        _lib.check(lib.tsq_join_build_push(h, bcols, 2, nb_local), h)
        _lib.check(lib.tsq_join_build_finish(h), h)
        ctx.sync()
        build_wall_ms = (time.time() - t0) * 1e3
        _lib.check(lib.tsq_join_set_count_only(h, 1), h)
        pcols = (abi.Col * 2)(dev_col(pk, npr), dev_col(pv, npr))

        def step():
            _lib.check(lib.tsq_join_probe_push(h, pcols, 2, npr, None), h)

        def full_sync():
            # single process, single stream: hipStreamSynchronize of the only stream with work
            ctx.sync()

    for _ in range(args.warmup):
        step()
    full_sync()
    if dj:
        dj.probed_local = dj.probe_batches = 0
    setup_s = time.time() - t_setup

    # ---------------------------------------------------------------- timed region: exactly K steps
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop_ms()  # HIP events on the stream the kernels were launched on
    full_sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        elapsed = comm.allreduce_f64([elapsed], parallel.Comm.MAX)[0]
    local_probe = [dj.probed_local, dj.probe_batches] if dj else [0, 0]  # rows this rank probed / probe batches (per-launch roofline)

    # N>1: one extra, untimed pass of the split + exchange alone (no probe), for SURVEY.md §8(d)'s t_exchange
    exchange_ms = None
    if distributed and not dj.shared:
        try:
            full_sync()
            te = time.perf_counter()
            parallel.redistribute_pieces(comm, pcols1, 0, 0, npr, args.exchange_chunks, lambda got, n: None)
            full_sync()
            exchange_ms = comm.allreduce_f64([time.perf_counter() - te], parallel.Comm.MAX)[0] * 1e3
        except Exception:  # reporting only
            exchange_ms = None

    # ---------------------------------------------------------------- verify (size-independent property)
    if distributed:
        total = dj.count()  # local counts, summed with one 8-byte all-reduce
    else:
        cnt = C.c_int64(0)
        _lib.check(lib.tsq_join_count(h, C.byref(cnt)), h)
        total = cnt.value
    # build keys are a bijection of [0, nb_global), every probe key lies in [0, nb_global): each probe row joins once
    expect = (args.steps + args.warmup) * npr * world
    if emu:
        # rank 0 alone holds build keys {(a i + b) mod M : i < nb}; a probe key k joins iff ((k - b) a^-1 mod M) < nb — numpy, on a host
        # copy of the probe keys (nothing of libtsq's join code computes it)
        import numpy as np
        host = np.empty(npr, dtype=np.int64)
        ctx.d2h(host, pk)
        ainv = pow(a_mult, -1, nb_global)
        idx = ((host - 12345) % nb_global).astype(np.uint64) * np.uint64(ainv) % np.uint64(nb_global)
        expect = (args.steps + args.warmup) * int(np.count_nonzero(idx < np.uint64(nb)))
        del host, idx
    ok = total == expect
    st = abi.Stats()
    _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
    # ... which a kernel that merely counted its rows would pass too.  N = 1: one more, untimed probe pass with keys uniform in
    # [0, 2 nb) (hit ratio 0.5, SURVEY.md §8d's rho variants) through the same handle; the expected count comes from numpy
    # on a host copy of that key column (nothing of libtsq's join code computes it).
    rho_check = None
    if not distributed:
        import numpy as np
        from tools.bench_sides import count_join_fracs
        pk2 = ctx.alloc(npr * 8)
        try:
            ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=5, m=2 * nb), npr, pk2)
            ctx.sync()
            host = np.empty(npr, dtype=np.int64)
            ctx.d2h(host, pk2)
            want_half = int(np.count_nonzero((host >= 0) & (host < nb)))
            del host
            pc2 = (abi.Col * 2)(dev_col(pk2, npr), dev_col(pv, npr))
            _lib.check(lib.tsq_join_probe_push(h, pc2, 2, npr, None), h)  # warm
            ctx.timer_start()
            _lib.check(lib.tsq_join_probe_push(h, pc2, 2, npr, None), h)
            half_ms = ctx.timer_stop_ms()
            c2 = C.c_int64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(c2)), h)
            got_half = (c2.value - total) // 2
            rho_check = {"workload": "the same join, probe keys uniform in [0, 2 N_b): hit ratio 0.5", "joined_rows": got_half, "expected_by_numpy": want_half,
                         "ms_per_probe_pass": half_ms, "rows_per_s": npr / half_ms * 1e3, **count_join_fracs(abi, st, npr, half_ms),
                         "verified": got_half == want_half and (c2.value - total) == 2 * want_half}
            ok = ok and rho_check["verified"]
        finally:
            ctx.free(pk2)
    plans = None
    if dj:
        # ---- N > 1: BOTH distributed plans in the one line.  `value` is the plan --dist-plan chose (shared images when
        # the build side packs: no probe row on xGMI); the other plan — north_star's hash-radix exchange with the RCCL all-to-all, or the
        # shared images when the exchange was asked for — runs the same K steps right after, between the same barriers, and is reported
        # beside it with its split / exchange / probe times.
        def plan_record(d, ms_step, verified, nsteps):
            sd = d.stats()
            rec = {"ms_per_step": ms_step, "rows_per_s": npr * world / (ms_step * 1e-3), "verified": bool(verified),
                   "local_probe_route": {abi.ROUTE_PACKED: "packed", abi.ROUTE_RADIX_LDS: "64-bit LDS", abi.ROUTE_RADIX_L2: "64-bit L2", abi.ROUTE_DIRECT: "direct"}.get(sd.probe_route, str(sd.probe_route))}
            if d.shared:
                rec.update({"wire_bytes_per_probe_row": 0.0, "allreduce_ms_once_per_build": sd.shared_allreduce_ms, "image_bytes": sd.shared_image_bytes})
            else:
                rec["wire_bytes_per_probe_row"] = 8.0 * (world - 1) / world
                if sd.radix_timed_batches > 0 and d.probe_batches > 0:  # local probe kernels of one step (a launch per received piece)
                    per_step = d.probe_batches / float(nsteps)
                    rec["t_probe_ms"] = (sd.partition_kernel_ms_sum + sd.radix_probe_kernel_ms_sum) / sd.radix_timed_batches * per_step
            return rec

        def exchange_times(rec):
            # t_partition: tsq_radix_split of this rank's probe keys alone; t_exchange: the split + all-to-all pass without probes, minus it
            try:
                spl = ctx.alloc(npr * 8)
                cnts = (C.c_int64 * max(world, 1))()
                one, dst = (abi.Col * 1)(dev_col(pk, npr)), (abi.Col * 1)(dev_col(spl, npr))
                _lib.check(lib.tsq_radix_split(ctx.h, one, 1, 0, 0, npr, world, dst, cnts), ctx.h)
                ctx.timer_start()
                for _ in range(3):
                    _lib.check(lib.tsq_radix_split(ctx.h, one, 1, 0, 0, npr, world, dst, cnts), ctx.h)
                rec["t_partition_ms"] = comm.allreduce_f64([ctx.timer_stop_ms() / 3.0], parallel.Comm.MAX)[0]
                ctx.free(spl)
                full_sync()
                te = time.perf_counter()
                parallel.redistribute_pieces(comm, pcols1, 0, 0, npr, args.exchange_chunks, lambda got, n: None)
                full_sync()
                both = comm.allreduce_f64([time.perf_counter() - te], parallel.Comm.MAX)[0] * 1e3
                rec["t_split_and_exchange_ms"] = both
                rec["t_exchange_ms"] = max(0.0, both - rec["t_partition_ms"])
            except Exception as e:  # reporting only
                rec["times_error"] = str(e)[:120]

        plans = {}
        main_key = "shared_images" if dj.shared else "hash_radix_exchange"
        plans[main_key] = plan_record(dj, elapsed / args.steps * 1e3, ok, args.steps)

        # The second plan is REPORTING: `value` above is measured and must reach the driver whatever happens next.  No N > 1 run on
        # hardware exists yet — a collective of the second plan that one rank never enters would hang every rank.  A watchdog thread
        # (ctypes calls release the GIL) gives the reporting part a deadline; past it rank 0 prints the line with what is known and
        # every rank leaves.
        def emit_what_is_known():
            if rank != 0:
                return
            plans["other_plan"] = {"error": "the second plan did not finish within its deadline; the line carries the chosen plan only"}
            line = {"metric": "probed rows/sec on int64-key inner hash join", "value": npr * world * args.steps / elapsed, "unit": "rows/s", "n_gpus": n_gpus,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                    "scaling": "strong" if args.rows_global > 0 else "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                    "config": {"workload": "SELECT count(*) FROM probe JOIN build ON k: %.0e x %.0e int64-key inner hash join per GPU, J-uniq-shuffled, hit ratio 1.0, "
                                           "build side resident in HBM" % (npr, nb), "probe_rows_per_gpu": npr, "build_rows_per_gpu": nb,
                               "parallelism": "%s x%d" % (main_key, world)},
                    "verified": bool(ok), "dist_plan": main_key, "plans": plans, "roofline": None, "cpu_baseline": None,
                    "note": "watchdog line: see plans.other_plan.error"}
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        disarm = arm_watchdog(max(120.0, 200.0 * elapsed), emit_what_is_known)
        if not dj.shared:
            exchange_times(plans[main_key])
        try:
            other = parallel.DistHashJoinCount(comm, cfg, radix_mode=radix_mode, packing_mode=abi.RADIX_OFF if args.packing == "off" else None, shared=not dj.shared)
            other.build([dev_col(bk, nb), dev_col(bv, nb)], 0, nb)
            okey = "shared_images" if other.shared else "hash_radix_exchange"
            if okey == main_key:
                plans["other_plan"] = {"available": False, "why": "the build side does not pack: tsq_join_build_finish_shared declined on every rank"}
            else:
                other.probe(pcols1, 0, npr, args.exchange_chunks)  # warm
                full_sync()
                c0 = other.count()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    other.probe(pcols1, 0, npr, args.exchange_chunks)
                full_sync()
                el2 = comm.allreduce_f64([time.perf_counter() - t1], parallel.Comm.MAX)[0]
                c1 = other.count()
                want2 = (expect // (args.steps + args.warmup)) * args.steps
                plans[okey] = plan_record(other, el2 / args.steps * 1e3, (c1 - c0) == want2, args.steps + 1)
                if not other.shared:
                    exchange_times(plans[okey])
            other.close()
        except Exception as e:  # reporting only: `value` stands
            plans["other_plan"] = {"error": str(e)[:160]}
        try:
            r_, n_, v_ = comm.info()
            plans["rccl"] = {"rank": r_, "nranks": n_, "version": v_}
        except Exception as e:
            plans["rccl"] = {"error": str(e)[:80]}
        disarm()
        dj.close()
    else:
        lib.tsq_join_destroy(h)

    rows_per_s = npr * world * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    step_ev_ms = ev_ms / args.steps  # HIP events around the K steps on the launch stream
    algo_bytes = 24.0 * npr
    wide = False
    radix = st.radix_batches > 0
    packed = st.probe_route == abi.ROUTE_PACKED
    part_name = "k_radix_partition<1024,16,4,0,false,true>"
    part_bytes_per_key = 16.0  # 8 B key read + 8 B table word written (COUNT(*) carries no payload)
    if radix and st.radix_timed_batches > 0:
        nt = min(st.radix_timed_batches, args.steps)  # the event ring keeps the most recent batches = the timed steps
        if packed:
            wide = st.packed_key_bits - st.radix_bits > 16
            kernel_name = "k_da_probe_count<1024,uint32_t>" if wide else "k_da_probe_count<512,uint16_t>"
            part_name = "k_da_partition<1024,16,uint32_t>" if wide else "k_da_partition2<512,8,4,true>"
            part_bytes_per_key = 8.0 + (4.0 if wide else 2.0)  # 8 B key read + one packed entry written
        else:
            kernel_name = "k_lds_probe_count<1024>" if st.table_slice_bits >= 3 else "k_radix_probe_count<2>"
        kernel_ms = st.radix_probe_kernel_ms_sum / st.radix_timed_batches
        part_ms = st.partition_kernel_ms_sum / st.radix_timed_batches
    else:
        nt = args.steps
        kernel_name = "k_probe_count<MULTI=0,GEN=0,CHK=0>"
        kernel_ms, part_ms = step_ev_ms, 0.0
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9

    out = {
        "knobs": args.knob or None,
        "metric": "probed rows/sec on int64-key inner hash join",
        "value": rows_per_s,
        "unit": "rows/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if args.rows_global > 0 else "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": "SELECT count(*) FROM probe JOIN build ON k: %.0e x %.0e int64-key inner hash join per GPU, "
                        "J-uniq-shuffled, hit ratio 1.0, build side resident in HBM%s" % (npr, nb, (
                            "; build key range %d bits <= 28: packed route (2-byte entries against direct-address images) — keys beyond 28 bits (31 if unique) "
                            "take the 64-bit route, see general_keys_64bit_route" % st.packed_key_bits) if packed else ""),
            "probe_rows_per_gpu": npr, "build_rows_per_gpu": nb,
            "parallelism": ("single GPU" if not distributed else
                            ("shared packed images x%d: every rank probes its own rows, the images are all-reduced once per build side (tsq_join_build_finish_shared)" % world
                             if dj.shared else "hash-radix x%d, RCCL send/recv all-to-all inside libtsq (tsq_redistribute)" % world)),
        },
        "verified": bool(ok),
        "joined_rows": total,
        "build_ms": build_wall_ms,
        "arena": {"reserved_gb": args.arena_gb, "reserve_s": t_arena,
                  "note": "build_ms is the FIRST build of the process; its buffers come out of the arena reserved at start-up "
                          "(tsq_ctx_reserve) — with --arena-gb 0 the same build pays ~35 ms per GB of first hipMalloc (310 ms before the arena existed, DESIGN.md §5)"},
        "build_kernel_ms": st.build_kernel_ms,
        "build_strategy": "partitioned: 2 radix passes + LDS slice images" if st.build_partitioned else "row-at-a-time CAS",
        "table_bytes": st.table_bytes,
        "setup_s": setup_s,
        "extras_file": args.extras_file,
    }
    if distributed:
        # what the plan puts on xGMI: per probe row in the timed step, and once per build side
        out["dist_plan"] = "shared_images" if dj.shared else "hash_radix_exchange"
        if emu:
            out["emulated_world"] = emu
            out["emulation_note"] = ("ONE GPU playing rank 0 of a %d-rank weak-scaling job: %d of the %d global build keys, images over the global %d-bit range; "
                                     "ms_per_step is the per-rank step of that job (the shared-images plan has no wire traffic in the step), so the "
                                     "projected aggregate is %d x value" % (emu, nb, nb_global, st.packed_key_bits, emu))
            out["projected_aggregate_rows_per_s"] = emu * rows_per_s if dj.shared else None
        out["wire_bytes_per_probe_row"] = 0.0 if dj.shared else 8.0 * (world - 1) / world
        if dj.shared:
            out["shared_images"] = {"image_bytes": st.shared_image_bytes, "allreduce_ms": st.shared_allreduce_ms,
                                    "wire_bytes_per_rank_once_per_build": 2.0 * st.shared_image_bytes * (world - 1) / world,
                                    "cells": "bit" if st.packed_key_bits > 28 else "byte", "key_range_bits": st.packed_key_bits}
    if plans:
        out["plans"] = plans
    if exchange_ms is not None:
        out["split_and_exchange_ms"] = exchange_ms  # tsq_redistribute of one step's probe keys (split + RCCL exchange), without the probes
    if packed:
        out["probe_strategy"] = ("packed keys: build-side key range of %d bits, 2^%d partitions of %d-byte entries against one-byte "
                                 "direct-address images of 2^%d cells in LDS" % (st.packed_key_bits, st.radix_bits, 4 if st.packed_key_bits - st.radix_bits > 16 else 2,
                                                                                   st.packed_key_bits - st.radix_bits))
        out["packed_images_ms"] = st.packed_build_ms
    else:
        out["probe_strategy"] = ("radix 2^%d partitions, table of 2^%d LDS-sized slices" % (st.radix_bits, st.table_slice_bits)) if radix else "direct"
    if rho_check is not None:
        out["rho_0.5"] = rho_check
    traffic = traffic_part = traffic_src = None
    for tf in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json"):  # PMC-derived HBM bytes per launch: measured offline (rocprofv3 --pmc passes), committed
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            w = tj["workload"]
            if w["probe_rows"] == npr and w["build_rows"] == nb and world == 1 and radix and w["radix_bits"] == st.radix_bits and kernel_name in tj and part_name in tj:
                traffic = tj[kernel_name]["traffic_bytes"]
                traffic_part = tj[part_name]["traffic_bytes"]
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950-corrected)" % tf
                break
        except Exception:
            pass
    if not distributed or dj.shared:  # (shared images: a rank's step IS the single-GPU step on its own rows)
        # `roofline` = the DOMINANT kernel of the step, priced at ITS OWN algorithmic bytes (contract: bytes per launch / launch duration);
        # `traffic_frac` = counter-measured HBM bytes / duration / peak — the fraction of the chip's bandwidth the kernel really uses.
        # `roofline.step` = the whole step (memsets + partition + probe + overflow kernels) the same two ways.  SURVEY.md 8(d) prices a
        # probe row at 24 B (8 B key + one 16 B slot); the packed route moves ~13 B per row, so that figure is kept only under its label.
        def frac_of(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 / 8000.0 if nbytes and ms > 0 else None
        pb = part_bytes_per_key * npr
        if packed:
            probe_algo = (4.0 if wide else 2.0) * npr + (((1 << st.packed_key_bits) / 8.0) if st.packed_key_bits > 28 else float(1 << st.packed_key_bits))
            probe_note = "entries read once + the direct-address images read once"
        else:
            probe_algo, probe_note = algo_bytes, "8 B table word + one 16 B slot per probe row (SURVEY.md 8d)"
        part = {"kernel": part_name, "kernel_ms": part_ms, "algorithmic_bytes_per_launch": pb, "achieved": pb / (part_ms * 1e-3) / 1e9 if part_ms > 0 else None,
                "frac": frac_of(pb, part_ms), "traffic": traffic_part, "traffic_frac": frac_of(traffic_part, part_ms)}
        probe_k = {"kernel": kernel_name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": probe_algo, "achieved": probe_algo / (kernel_ms * 1e-3) / 1e9,
                   "frac": frac_of(probe_algo, kernel_ms), "traffic": traffic, "traffic_frac": frac_of(traffic, kernel_ms), "note": probe_note}
        step_traffic = (traffic + traffic_part) if traffic and traffic_part else None
        step = {"ms": step_ev_ms, "algorithmic_bytes": pb + probe_algo if radix else algo_bytes, "frac": frac_of(pb + probe_algo if radix else algo_bytes, step_ev_ms),
                "traffic": step_traffic, "traffic_frac": frac_of(step_traffic, step_ev_ms),
                "frac_priced_at_24B_per_probe_row": frac_of(algo_bytes, step_ev_ms),
                "note": "whole step: memsets + partition + probe + overflow kernels.  `frac` = the bytes THIS algorithm must move (per-kernel algorithmic bytes "
                        "summed) / step time / 8 TB/s; `traffic_frac` = counter-measured HBM bytes / step time / 8 TB/s; `frac_priced_at_24B_per_probe_row` = "
                        "SURVEY.md 8(d)'s pricing (8 B key + one 16 B slot per probe row), the north star's probe-phase yardstick — it exceeds the bytes the "
                        "packed route moves (~13 B per row), so it measures progress against the 64-bit design, not bandwidth use"}
        dom, other = (part, probe_k) if (radix and part_ms > kernel_ms) else (probe_k, part)
        out["roofline"] = {"bound": "hbm", "achieved": dom["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
                           "traffic_frac": dom["traffic_frac"], "traffic_source": traffic_src, "kernel": dom["kernel"], "kernel_ms": dom["kernel_ms"],
                           "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "step": step, "probe_phase": step}
        if radix:
            out["roofline"]["second_kernel"] = other
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
            "partition": {"kernel": part_name, "kernel_ms": part_ms,
                          "algorithmic_bytes_per_launch": part_bytes_per_key * rows_per_launch,
                          "achieved": part_bytes_per_key * rows_per_launch / (part_ms * 1e-3) / 1e9 if part_ms > 0 else None},
        }

    # ---------------------------------------------------------------- side measurements (N = 1): tools/bench_sides.py
    if not distributed and not args.no_extras and nb == 100_000_000 and npr == 100_000_000:
        from tools import bench_sides as S
        want = [k for k in args.only_extras.split(",") if k]
        for key, fn in S.registry(ctx, abi, _lib, bk, bv, pk, pv, nb, npr):
            if want and key not in want:
                continue
            try:
                out[key] = fn()
            except Exception as e:  # reporting only
                out[key] = {"error": str(e)[:200]}
        S.attach_counter_traffic(out)

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(abi, args.cpu_build_rows, args.cpu_probe_rows)
        except Exception as e:  # the baseline is reporting only; never fail the bench for it
            out["cpu_baseline"] = {"error": str(e)[:200]}

    if "arena" in out:
        out["arena"]["peak_gb"] = ctx.arena_stats()["peak"] / float(1 << 30)
    for p in (bk, bv, pk, pv):
        ctx.free(p)
    if comm:
        comm.close()
    ctx.close()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)  # C stdio too (RCCL announces itself with printf: on a pipe that text would come out at exit, after the line)
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        # the whole record (every side measurement with its workload, check and counters) goes to a side file; stdout gets ONE line < 4 KB
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", args.extras_file), "w") as f:
                json.dump(out, f, indent=1)
            sys.stderr.write("bench: full record (%d keys) in gpurun_out/%s\n" % (len(out), args.extras_file))
        except OSError as e:
            sys.stderr.write("bench: could not write the extras file: %s\n" % e)
        sys.stderr.flush()
        print(compact_line(out))
        sys.stdout.flush()
    if not ok:
        sys.exit("bench: join count mismatch: got %d expected %d" % (total, expect))


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
LINE_LIMIT = 4096  # bytes of the ONE stdout line (a 21 KB line was not parsed by the driver)


def _r(x, sig=4):
    """numbers of the line keep `sig` significant digits (ints and everything else pass)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.*g" % (sig, x))


def _side_entry(v):
    """one side measurement in the line: {ms, frac, ok} (+ rows_per_s where there is no ms).  `frac` = the bytes the route must move /
    time / 8 TB/s (tools/bench_sides.py count_join_fracs); the SURVEY 8(d)-priced figure stays in the extras file as frac_8d."""
    e = {}
    ms = v.get("ms", v.get("ms_per_probe_pass"))
    if ms is not None:
        e["ms"] = _r(ms)
    elif "probe_rows_per_s_end_to_end" in v:
        e["rows_per_s"] = _r(v["probe_rows_per_s_end_to_end"])
    elif "build_ms_warm" in v:
        e["ms"] = _r(v["build_ms_warm"])
    if v.get("frac") is not None:
        e["frac"] = _r(v["frac"], 3)
    if "error" in v:
        e["error"] = str(v["error"])[:60]
    elif isinstance(v.get("verified"), bool):
        e["ok"] = v["verified"]
    return e


def compact_line(full, limit=LINE_LIMIT):
    """the ONE stdout line: the contract keys, `config`, `roofline` (dominant kernel + whole step), `cpu_baseline` and {ms, frac, ok} per
    side measurement — strict JSON, shorter than `limit` bytes whatever the side measurements returned."""
    line = {k: _r(full.get(k), 6) for k in CONTRACT_KEYS}
    line["config"] = full.get("config")
    line["verified"] = full.get("verified")
    wk = full.get("wide_keys_64bit_route")
    if isinstance(wk, dict) and wk.get("rows_per_s") is not None:  # the same join with arbitrary int64 keys: never hidden behind the packed headline
        line["general_keys_64bit_route"] = {"value": _r(wk["rows_per_s"]), "ms_per_step": _r(wk.get("ms_per_probe_pass")), "frac_at_24B_per_row": _r(wk.get("frac"), 3), "ok": wk.get("verified")}
    for k in ("dist_plan", "emulated_world", "wire_bytes_per_probe_row", "split_and_exchange_ms", "build_ms"):
        if full.get(k) is not None:
            line[k] = _r(full[k])
    if isinstance(full.get("shared_images"), dict):
        line["shared_images"] = {k: _r(v) for k, v in full["shared_images"].items() if k in ("image_bytes", "allreduce_ms", "cells", "key_range_bits")}
    if isinstance(full.get("plans"), dict):
        keep = ("ms_per_step", "rows_per_s", "verified", "t_partition_ms", "t_exchange_ms", "t_probe_ms", "wire_bytes_per_probe_row", "allreduce_ms_once_per_build",
                "available", "error", "rank", "nranks", "version")
        line["plans"] = {k: {kk: (_r(vv) if not isinstance(vv, str) else vv[:60]) for kk, vv in v.items() if kk in keep} for k, v in full["plans"].items() if isinstance(v, dict)}
    rf = full.get("roofline")
    if isinstance(rf, dict):
        r = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac", "kernel", "kernel_ms") if k in rf}
        if isinstance(rf.get("step"), dict):
            r["step"] = {k: _r(rf["step"].get(k)) for k in ("ms", "frac", "traffic_frac", "frac_priced_at_24B_per_probe_row")}
        if isinstance(rf.get("second_kernel"), dict):
            r["second_kernel"] = {k: _r(rf["second_kernel"].get(k)) for k in ("kernel", "kernel_ms", "frac", "traffic_frac")}
        line["roofline"] = r
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: (_r(v) if not isinstance(v, str) else v[:120]) for k, v in cb.items() if k in ("value", "unit", "cores", "kind", "sample", "error")}
    sides = {}
    for k, v in full.items():
        if k in ("config", "roofline", "cpu_baseline", "arena", "shared_images", "plans") or not isinstance(v, dict):
            continue
        subs = {kk: vv for kk, vv in v.items() if isinstance(vv, dict) and ("verified" in vv or "error" in vv)}
        if ("verified" in v or "error" in v or "ms" in v or "build_ms_warm" in v) and not (subs and "ms" not in v and "ms_per_probe_pass" not in v):
            sides[k] = _side_entry(v)
        for kk, vv in subs.items():
            if "borrowed_pulls" in kk:  # (a variant of the boundary measurement that measured equal: in the extras file, not in the line)
                continue
            sides["%s.%s" % (k, kk)] = _side_entry(vv)
    if sides:
        line["sides"] = sides
    line["extras_file"] = "gpurun_out/" + str(full.get("extras_file", "bench_extras.json"))
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # never longer than the limit: first the side entries lose their numbers (ok stays), then the sides go, then the free text shrinks
    if len(text) >= limit and sides:
        line["sides"] = {k: ({"ok": e["ok"]} if "ok" in e else {"error": 1} if "error" in e else {}) for k, e in sides.items()}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= limit and sides:
        line["sides"] = {"dropped": len(sides), "all_ok": all(e.get("ok", True) and "error" not in e for e in sides.values())}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= limit:
        if isinstance(line.get("config"), dict):
            line["config"] = {k: (v[:100] if isinstance(v, str) else v) for k, v in line["config"].items()}
        if isinstance(line.get("cpu_baseline"), dict) and isinstance(line["cpu_baseline"].get("sample"), str):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:40]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < limit, len(text)
    return text


def self_launch(args):
    """`python3 bench.py --gpus N` without a launcher: spawn the N ranks (one process per GPU) ourselves, with the environment
    torch.distributed.run would give them; rank 0 inherits stdout (the ONE line), the other ranks' stdout goes to stderr."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        for pr in procs:
            rc = max(rc, abs(pr.wait()))
    finally:
        for pr in procs:  # a rank died: the others would wait for it in a collective for ever
            if pr.poll() is None:
                pr.kill()
    sys.exit(rc)


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
    # the reference's worker counts: executor/benchmark_test.go:357 runs 4 join workers, tidb_hash_join_concurrency defaults to 5
    # (sessionctx/variable/tidb_vars.go:249) — the reported `value` — and, as the most the host can do, one worker per core
    counts = sorted({min(cores, 4), min(cores, 5), cores})
    n, build_ms, probe_ms = orc.hash_join_timed_multi(cfg, build, probe, counts)
    assert n == npr
    by = {c: npr / (ms * 1e-3) for c, ms in zip(counts, probe_ms)}
    threads = min(cores, 5)
    return {
        "value": by[threads], "unit": "rows/s", "cores": threads, "kind": "port",
        "sample": "C++ restatement of the reference HashJoinExec algorithm (FNV-1 + chained map, 1024-row chunks), "
                  "%d probe worker threads of %d host cores, %.0e probe rows x %.0e build rows of the same generators; "
                  "build %.0f ms single-threaded (once), probe %.0f ms" % (threads, cores, npr, nb, build_ms, probe_ms[counts.index(threads)]),
        "build_rows_per_s": nb / (build_ms * 1e-3),
        "rows_per_s_by_probe_threads": {str(c): by[c] for c in counts},
    }


if __name__ == "__main__":
    main()
