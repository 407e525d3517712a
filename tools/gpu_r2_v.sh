#!/bin/bash
# coprocessor chain (order-preserving compaction) + the tests that use tsq_chunk_compact + bench with the 32-bit bucket arithmetic in the LDS probe
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2v}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_storage_coprocessor_gpu.py tests/test_expr_gpu.py tests/test_pipeline_gpu.py tests/test_join_radix_gpu.py tests/test_join_varlen_gpu.py tests/test_storage_rows_encode_gpu.py -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P' $O/bench.json
import json,sys
d=json.loads(open(sys.argv[1]).read())
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['probe_phase']['frac'], d['roofline']['partition']['kernel_ms'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified')})
P
