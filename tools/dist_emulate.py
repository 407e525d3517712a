#!/usr/bin/env python3
"""One GPU playing rank 0 of a W-rank weak-scaling job (BASELINE configs[3] shape: COUNT(*) of an int64-key inner join, N_b build and
N_p probe rows PER RANK) under BOTH distributed plans, for W = 2, 4, 8 — what the first hardware SCALE run will execute, measured
kernel by kernel where a single GPU can measure it and priced where it cannot.

  shared images  : bench.py --force-dist --emulate-world W (rank 0's 1 / W of the global build keys, images over the global range, the
                   rank's own probe rows; nothing on xGMI inside the step).  The all-reduce of the images happens once per build side:
                   priced here as a ring all-reduce, 2 (W - 1) / W x image bytes over one xGMI link (153 GB/s) — reported, and ADDED to
                   the step for the one-pass shape (configs[3] probes its build side once).
  hash-radix     : the fallback when the build keys do not pack.  Rank 0 after the exchange holds the build rows whose keys rank to it;
  exchange         per step it (a) splits its own N_p probe rows by rank(key) into W runs (tsq_radix_split: measured), (b) sends (W - 1) / W
                   of them and receives as many (PRICED: every peer pair has its own xGMI link, so the all-to-all takes N_p x 8 / W bytes
                   over one 153 GB/s link each way), (c) probes the N_p rows it ends up with against its local table (measured: the
                   received rows are built by splitting the GLOBAL probe side and keeping run 0).
Every probe key of the global domain joins exactly one build row (the build keys are a bijection of [0, W N_b)), so the expected
count of a step is the number of probe rows that rank to 0 — checked.

usage: dist_emulate.py [rows_per_rank] -> ONE JSON object on stdout (profiles/r05_dist_emulated.json)"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402

XGMI_LINK_GBS = 153.0  # per peer link and direction (MI355X_MICROARCH.md; 7 links per GPU)
A_MULT = 2654435761


def dev_col(ptr, n):
    c = abi.Col()
    c.data, c.length, c.elem_size, c.type, c.flags = ptr, n, 8, abi.I64, abi.COL_DEVICE
    return c


def spec(kind, **kw):
    s = abi.GenSpec()
    s.kind, s.seed = kind, 42
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def exchange_plan(ctx, n, W, steps=5, packing=True):
    lib = ctx.lib
    M = n * W
    res = {"world": W, "rows_per_rank": n}
    gb, gp, sb, sp_, own, own_split = (ctx.alloc(M * 8), ctx.alloc(M * 8), ctx.alloc(M * 8), ctx.alloc(M * 8), ctx.alloc(n * 8), ctx.alloc(n * 8))
    try:
        ctx.gen_column(spec(abi.GEN_AFFINE, table=2, a=A_MULT, b=12345, m=M), M, gb)       # the GLOBAL build side: a bijection of [0, M)
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=M), M, gp)                  # the GLOBAL probe side
        ctx.gen_column(spec(abi.GEN_RAND_MOD, table=1, col=0, m=M), n, own)                 # rank 0's own probe rows (rows [0, n) of it)
        cb, cp = (C.c_int64 * W)(), (C.c_int64 * W)()
        _lib.check(lib.tsq_radix_split(ctx.h, (abi.Col * 1)(dev_col(gb, M)), 1, 0, 0, M, W, (abi.Col * 1)(dev_col(sb, M)), cb), ctx.h)
        _lib.check(lib.tsq_radix_split(ctx.h, (abi.Col * 1)(dev_col(gp, M)), 1, 0, 0, M, W, (abi.Col * 1)(dev_col(sp_, M)), cp), ctx.h)
        nb0, np0 = int(cb[0]), int(cp[0])  # run 0 of each = what rank 0 holds after the exchange
        res["build_rows_on_rank0"], res["probe_rows_on_rank0"] = nb0, np0
        # (a) the send side of a step: rank 0's own rows split into W runs
        co = (C.c_int64 * W)()
        split = lambda: _lib.check(lib.tsq_radix_split(ctx.h, (abi.Col * 1)(dev_col(own, n)), 1, 0, 0, n, W, (abi.Col * 1)(dev_col(own_split, n)), co), ctx.h)  # noqa: E731
        split()
        ctx.sync()
        ctx.timer_start()
        for _ in range(steps):
            split()
        res["split_ms"] = ctx.timer_stop_ms() / steps
        res["rows_kept_of_own"] = int(co[0])
        # (c) the local join: build = run 0 of the global build side, probe = run 0 of the global probe side
        cfg = abi.JoinCfg()
        cfg.join_type, cfg.build_is_right, cfg.n_keys, cfg.n_build_cols, cfg.n_probe_cols = abi.JOIN_INNER, 1, 1, 1, 1
        cfg.build_types[0] = cfg.probe_types[0] = abi.I64
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            if not packing:
                _lib.check(lib.tsq_join_set_key_packing(h, abi.RADIX_OFF), h)
            _lib.check(lib.tsq_join_build_push(h, (abi.Col * 1)(dev_col(sb, nb0)), 1, nb0), h)
            _lib.check(lib.tsq_join_build_finish(h), h)
            _lib.check(lib.tsq_join_set_count_only(h, 1), h)
            pc = (abi.Col * 1)(dev_col(sp_, np0))
            _lib.check(lib.tsq_join_probe_push(h, pc, 1, np0, None), h)
            ctx.sync()
            ctx.timer_start()
            for _ in range(steps):
                _lib.check(lib.tsq_join_probe_push(h, pc, 1, np0, None), h)
            res["probe_ms"] = ctx.timer_stop_ms() / steps
            cnt = C.c_int64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(cnt)), h)
            st = abi.Stats()
            _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
            res["verified"] = cnt.value == (steps + 1) * np0
            res["probe_route"] = int(st.probe_route)
            res["packed_key_bits"] = int(st.packed_key_bits)
        finally:
            lib.tsq_join_destroy(h)
    finally:
        for p in (gb, gp, sb, sp_, own, own_split):
            ctx.free(p)
    wire_bytes = n * 8.0 * (W - 1) / W                       # sent by rank 0 per step (and received)
    res["wire_bytes_per_step_each_way"] = wire_bytes
    res["wire_ms_priced"] = (n * 8.0 / W) / (XGMI_LINK_GBS * 1e9) * 1e3  # every peer has its own link: the slowest pair moves n 8 / W bytes
    serial = res["split_ms"] + res["wire_ms_priced"] + res["probe_ms"]
    overlapped = res["split_ms"] + max(res["wire_ms_priced"], res["probe_ms"])   # bench.py --exchange-chunks: piece c + 1 on the wire while c is probed
    res["step_ms_serial"], res["step_ms_overlapped"] = serial, overlapped
    res["aggregate_rows_per_s_overlapped"] = W * n / overlapped * 1e3
    return res


def shared_plan(n, W, steps=10):
    """bench.py's own emulation of the shared-images plan (rank 0 of W), in a process of its own"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--force-dist", "--steps", str(steps), "--warmup", "2",
           "--build-rows", str(n), "--probe-rows", str(n), "--extras-file", "dist_emulate_tmp.json"]
    if W > 1:
        cmd += ["--emulate-world", str(W)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"world": W, "error": (r.stderr or r.stdout)[-300:]}
    d = json.loads(line[-1])
    img = (d.get("shared_images") or {})
    out = {"world": W, "rows_per_rank": n, "step_ms": d["ms_per_step"], "verified": d["verified"], "image_bytes": img.get("image_bytes"),
           "cells": img.get("cells"), "key_range_bits": img.get("key_range_bits"), "wire_bytes_per_probe_row": d.get("wire_bytes_per_probe_row")}
    if img.get("image_bytes"):
        # ring all-reduce of the images, once per build side: 2 (W - 1) / W x bytes through one link
        out["allreduce_ms_priced_once_per_build"] = 2.0 * (W - 1) / W * img["image_bytes"] / (XGMI_LINK_GBS * 1e9) * 1e3 if W > 1 else 0.0
        out["aggregate_rows_per_s_repeated_probes"] = W * n / d["ms_per_step"] * 1e3
        one = d["ms_per_step"] + out["allreduce_ms_priced_once_per_build"]
        out["step_ms_one_pass_incl_allreduce"] = one
        out["aggregate_rows_per_s_one_pass"] = W * n / one * 1e3
    return out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    out = {"what": __doc__.split("\n\n")[0], "rows_per_rank": n, "xgmi_link_GBs": XGMI_LINK_GBS, "shared_images": {}, "exchange_packed": {}, "exchange_64bit": {}}
    for W in (1, 2, 4, 8):
        out["shared_images"]["W%d" % W] = shared_plan(n, W)
    with _lib.Context(0) as ctx:
        ctx.reserve(96 << 30)
        for W in (2, 4, 8):
            out["exchange_packed"]["W%d" % W] = exchange_plan(ctx, n, W)
            out["exchange_64bit"]["W%d" % W] = exchange_plan(ctx, n, W, packing=False)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
