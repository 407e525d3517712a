#!/bin/bash
# validation of HEAD: full GPU parity suite, the default bench line, rocprofv3 kernel stats of the bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2t}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof.json 2> $R/$O/prof.err
cd $R
python tools/summarize_prof.py $O/rocprof.txt --stats $O/prof/p_kernel_stats.csv --note "python bench.py --steps 20 --warmup 5 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (HEAD of round 2, session 2)"
head -14 $O/rocprof.txt | cut -c1-150
python - <<'P' $O/bench.json
import json,sys
d=json.loads(open(sys.argv[1]).read())
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['probe_phase']['frac'], d['roofline']['partition']['kernel_ms'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified')})
print(d.get('cpu_baseline'))
P
