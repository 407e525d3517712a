"""CPU: the bench line committed under profiles/ keeps the driver's contract (the keys bench.py must print, the `roofline` and
`cpu_baseline` objects of the task's measurement section), every side measurement in it carried a passed check when it was taken,
and profiles/traffic_r04.json is what tools/make_traffic.py makes of the committed PMC summaries (the file bench.py reads its
counter traffic from)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r04_bench.json")).read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "int64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["probe_rows_per_gpu"] / d["ms_per_step"] * 1e3) / d["value"] < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] < 1.0 and 0.0 < r["traffic_frac"] < 1.0 and 0.0 < r["step"]["traffic_frac"] < 1.0  # no unlabelled fraction above 1
    assert "traffic_r04.json" in r["traffic_source"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 5 and set(c["rows_per_s_by_probe_threads"]) >= {"4", "5"}


def test_every_side_measurement_of_the_line_was_verified():
    d = _line()
    seen = 0
    for k, v in d.items():
        if not isinstance(v, dict):
            continue
        assert "error" not in v, (k, v.get("error"))
        if "verified" in v:
            seen += 1
            assert v["verified"], k
        for kk, vv in v.items():
            if isinstance(vv, dict) and "verified" in vv:
                seen += 1
                assert vv["verified"], (k, kk)
    assert seen >= 20 and d["verified"] is True
    for k in ("c2_1e8x1e7", "c3_agg_1e9_1e6", "c3_zipf_s1", "c3_sparse_keys", "q3_sf100", "materialising", "two_key_columns_count_48bit",
              "two_key_bigint_string_count", "variants_8d", "pcie_inclusive_1e7", "wide_keys_64bit_route"):
        assert k in d, k


import pytest


@pytest.mark.parametrize("rnd", ["r04", "r05"])
def test_traffic_file_is_what_the_tool_makes_of_the_committed_pmc_summary(tmp_path, rnd):
    out = tmp_path / "t.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), os.path.join(ROOT, "profiles", "%s_bench_pmc.txt" % rnd), str(out)],
                   check=True, capture_output=True)
    made, have = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", "traffic_%s.json" % rnd)))
    for k in ("k_da_partition2<512,8,4,true>", "k_da_probe_count<512,uint16_t>", "workload", "kernels_KiB_per_launch"):
        assert made[k] == have[k], k
    if rnd == "r04":
        assert _line()["roofline"]["traffic"] == have["k_da_partition2<512,8,4,true>"]["traffic_bytes"]
    # counter traffic of the dominant kernel within 10 % of the bytes the algorithm must move (1e8 x (8 B key + 2 B entry)): no wasted re-reads
    assert 0.9 < have["k_da_partition2<512,8,4,true>"]["traffic_bytes"] / 1.0e9 < 1.1


# ---------------------------------------------------------------- round 5: the ONE stdout line is small and strict JSON
def _bench_module():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _strict(text):
    def bad(c):
        raise ValueError("non-strict JSON constant " + c)
    return json.loads(text, parse_constant=bad)


def test_the_stdout_line_of_a_full_record_is_under_4_KB_and_strict_json():
    bench = _bench_module()
    full = _line()  # round 4's 21 KB record = what main() hands to compact_line()
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full)
    assert "\n" not in text and len(text.encode()) < 4096
    d = _strict(text)
    for k in bench.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline", "sides", "verified"):
        assert k in d, k
    assert d["config"]["workload"].startswith("SELECT count(*)") and d["dtype"] == "int64" and d["n_gpus"] == 1
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac", "kernel", "kernel_ms", "step"):
        assert k in r, k
    assert set(r["step"]) == {"ms", "frac", "traffic_frac", "frac_priced_at_24B_per_probe_row"} and "probe_phase" not in r
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    # every side measurement: at most {ms | rows_per_s, frac, ok | error}
    assert len(d["sides"]) >= 20
    for k, e in d["sides"].items():
        assert set(e) <= {"ms", "rows_per_s", "frac", "ok", "error"}, (k, e)
    for k in ("c2_1e8x1e7", "c3_agg_1e9_1e6", "wide_keys_64bit_route", "materialising", "q3_sf100", "variants_8d.rho_0.1", "pcie_inclusive_1e7.chunks_of_1024_rows"):
        assert k in d["sides"], k


def test_the_stdout_line_stays_under_the_limit_whatever_the_sides_return():
    bench = _bench_module()
    full = _line()
    for i in range(200):  # many more side measurements than fit, each with paragraphs of prose
        full["side_%03d" % i] = {"ms": 1.0 / 3.0, "frac": 2.0 / 3.0, "verified": i != 7, "workload": "x" * 500, "note": {"a": ["y" * 100] * 10}}
    full["config"]["parallelism"] = "p" * 3000
    full["cpu_baseline"]["sample"] = "s" * 3000
    text = bench.compact_line(full)
    assert len(text.encode()) < 4096
    d = _strict(text)
    assert d["value"] == full["value"] or abs(d["value"] - full["value"]) / full["value"] < 1e-5
    assert d["sides"] == {"dropped": 223, "all_ok": False} or "side_007" in d["sides"]
    # NaN / inf never reach the line (strict JSON): compact_line refuses them
    full2 = _line()
    full2["ms_per_step"] = float("nan")
    try:
        bench.compact_line(full2)
        raised = False
    except ValueError:
        raised = True
    assert raised


def test_bench_gpus_n_without_a_launcher_spawns_its_ranks(tmp_path):
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment must become the launcher: checked
    here without a GPU by letting the ranks fail at context creation — each must have been started with its own RANK / WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["TSQ_BENCH_ECHO_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    seen = sorted(l for l in (r.stdout + r.stderr).splitlines() if l.startswith("bench-rank "))
    assert seen == ["bench-rank 0 of 2 local 0", "bench-rank 1 of 2 local 1"], (r.stdout[-500:], r.stderr[-500:])


# ---- round 5: the line the driver's command printed on the GPU box (`tools/gpu.sh bench --gpus 1 --steps 20 --warmup 5`), in the compact form (< 4 KB) it parses
def _line5():
    raw = open(os.path.join(ROOT, "profiles", "r05_bench.json")).read().strip().splitlines()
    assert len(raw) == 1, "stdout of bench.py is ONE line"
    assert len(raw[0]) < 4096
    return json.loads(raw[0])


def test_round5_line_has_the_contract_keys_roofline_and_cpu_baseline():
    d = _line5()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5)  # the driver's command
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "int64" and d["verified"] is True
    assert "1e+08 x 1e+08" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["probe_rows_per_gpu"] / d["ms_per_step"] * 1e3) / d["value"] < 0.02
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.4 < r["frac"] < 1.0 and 0.0 < r["traffic_frac"] < 1.0 and 0.0 < r["step"]["frac"] < 1.0 and r["traffic"] > 5e8
    assert r["kernel_ms"] < r["step"]["ms"] <= d["ms_per_step"] * 1.05  # the dominant kernel fits its step, the step the wall clock
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 5 and c["unit"] == "rows/s" and 1e6 < c["value"] < 1e8 and c["sample"]


def test_round5_every_side_measurement_passed_its_check():
    s = _line5()["sides"]
    assert len(s) >= 30
    for k, e in s.items():
        assert "error" not in e and e.get("ok", True) is True, k
    for k in ("c2_1e8x1e7", "c3_agg_1e9_1e6", "c3_zipf_s1", "c3_sparse_keys", "q3_sf100", "materialising", "two_key_bigint_string_count", "two_key_bigint_string_count.rows",
              "wide_keys_64bit_route", "stream_agg_1e8_ordered", "agg_string_keys_1e7_1e5", "agg_string_keys_1e7_5e6", "expr_kernels.arith_int64",
              "pcie_inclusive_1e7.native_chunks_of_1024_rows"):
        assert k in s, k


# ---------------------------------------------------------------- round 6: the roofline is reproducible from profiles/ 
def _line6():
    raw = [l for l in open(os.path.join(ROOT, "profiles", "r06_bench.json")) if l.startswith("{")]
    assert len(raw) == 1 and len(raw[0]) < 4096  # ONE line on stdout, as the driver's command printed it on the GPU box
    return _strict(raw[0])


def _rocprof_avg_us(path, stem):
    for l in open(path):
        if l.startswith(stem):
            f = l.split()
            return float(f[-2]), int(f[-4])  # avg_us, calls
    raise AssertionError("%s: no row for %s" % (path, stem))


def test_round6_line_keeps_the_contract_and_names_the_general_key_number():
    d = _line6()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5) and d["verified"] is True and d["vs_baseline"] is None
    assert "packed route" in d["config"]["workload"] and "28" in d["config"]["workload"]  # the headline's key range is named ...
    g = d["general_keys_64bit_route"]                                                      # ... and the number for any 64-bit keys stands next to `value`
    assert g["ok"] is True and 0 < g["value"] < d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.4 < r["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_round6_kernel_ms_of_the_line_agrees_with_the_committed_rocprof_summary():
    # profiles/r06_headline_rocprof.txt = rocprofv3 --kernel-trace --stats of `bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5`
    # (the headline's dispatches only); bench.py's own HIP-event averages of the same kernels must agree with it within 5 %
    r = _line6()["roofline"]
    prof = os.path.join(ROOT, "profiles", "r06_headline_rocprof.txt")
    avg, calls = _rocprof_avg_us(prof, "void k_da_partition2<512, 8, 4, true, unsigned short, false, false>")
    assert calls >= 25 and abs(r["kernel_ms"] * 1e3 - avg) / avg < 0.05, (r["kernel_ms"], avg)
    avg2, _ = _rocprof_avg_us(prof, "void k_da_probe_count<512, unsigned short, false, false, false>")
    assert abs(r["second_kernel"]["kernel_ms"] * 1e3 - avg2) / avg2 < 0.10, (r["second_kernel"]["kernel_ms"], avg2)
    assert r["kernel_ms"] + r["second_kernel"]["kernel_ms"] <= r["step"]["ms"] * 1.02


def test_round6_traffic_file_is_what_the_tool_makes_of_the_headline_pmc_passes(tmp_path):
    out = tmp_path / "t.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), os.path.join(ROOT, "profiles", "r06_headline_pmc.txt"), str(out)], check=True, capture_output=True)
    made, have = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", "traffic_r06.json")))
    for k in ("k_da_partition2<512,8,4,true>", "k_da_probe_count<512,uint16_t>", "workload"):
        assert made[k] == have[k], k
    p = have["k_da_partition2<512,8,4,true>"]
    # reads: the 8-byte keys once (2 x FETCH_SIZE, the gfx950 correction); writes: 2-byte entries in 16-byte runs — the counters see up to
    # 1.6x the 0.2 GB of entries (partial lines leaving the L2 more than once); no re-reads
    assert 0.98 < 2 * p["FETCH_SIZE_KiB"] * 1024 / 0.8e9 < 1.08 and 0.9 < p["WRITE_SIZE_KiB"] * 1024 / 0.2e9 < 1.8
    assert p["launches"] >= 25


def test_round6_sides_carry_the_reference_benchmark_shapes_and_the_targets_met():
    s = _line6()["sides"]
    for k, e in s.items():
        assert "error" not in e and e.get("ok", True) is True, k
    for k in ("ref_BenchmarkHashJoinExec_keyIdx01", "ref_BenchmarkHashJoinExec_keyIdx0", "ref_BenchmarkAggRows_1e7_ndv1000", "ref_BenchmarkAggNDV_1e7_ndv1e7",
              "materialising", "materialising_nullable_left_outer", "q3_sf100", "c3_agg_1e9_1e6", "c3_agg_1e9_1e6_double", "expr_kernels.arith_int64"):
        assert k in s, k
    assert s["materialising"]["ms"] <= 4.5 and s["materialising_nullable_left_outer"]["ms"] <= 7.0  # VERDICT r5 item 1's one-pass targets


def test_watchdog_prints_what_is_known_and_leaves_when_the_reporting_part_hangs():
    # bench.py at N > 1 measures the second distributed plan after `value` is known; a collective that hangs there must not cost the line
    code = ("import sys, time, json; sys.path.insert(0, %r); import bench\n"
            "disarm = bench.arm_watchdog(0.3, lambda: (sys.stdout.write(json.dumps({'value': 1.5, 'plans': {'other_plan': {'error': 'deadline'}}}) + '\\n'), sys.stdout.flush()))\n"
            "time.sleep(30)\nprint('not reached')\n") % ROOT
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert time.time() - t0 < 20 and r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["value"] == 1.5 and "not reached" not in r.stdout
    # disarmed in time: nothing is printed, the program goes on
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "disarm = bench.arm_watchdog(0.3, lambda: print('late'))\ndisarm()\ntime.sleep(0.8)\nprint('went on')\n") % ROOT
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=25)
    assert r2.returncode == 0 and "went on" in r2.stdout and "late" not in r2.stdout
