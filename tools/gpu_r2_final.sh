#!/bin/bash
# end-of-session validation: full GPU parity suite, smoke, the default bench line, rocprofv3 stats of the bench workload, Q3 pipeline, build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2final}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 1200 python -m pytest tests -q -m gpu --durations=12 > $O/pytest.txt 2>&1
tail -18 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/$O/prof.json 2> $R/$O/prof.err
cd $R
python tools/summarize_prof.py $O/rocprof_noextras.txt --stats $O/prof/p_kernel_stats.csv --note "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras under rocprofv3 --kernel-trace --stats (end of round 2, session 2)"
head -9 $O/rocprof_noextras.txt | cut -c1-150
python - <<'P' $O/bench.json
import json,sys
d=json.loads(open(sys.argv[1]).read())
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['probe_phase']['frac'], d['roofline']['partition']['kernel_ms'], d['build_kernel_ms'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified')})
P
timeout 300 python tools/q3.py 10 > $O/q3.txt 2>&1; tail -2 $O/q3.txt | cut -c1-400
timeout 300 python tools/q3.py 10 --topn > $O/q3_topn.txt 2>&1; tail -1 $O/q3_topn.txt | cut -c1-300
