#!/bin/bash
# 2^12 probe partitions (16-bit packed cursors): parity, then the bench with and without
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2f2}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_join_radix_gpu.py -q -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for pb in 1 0; do
  TSQ_RADIX_PB12=$pb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_pb12_$pb.json 2> $O/bench_pb12_$pb.err
  python - <<P $O/bench_pb12_$pb.json
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("PB12=$pb", {k:d[k] for k in ('ms_per_step','verified','probe_strategy')}, d['roofline']['kernel_ms'], d['roofline']['probe_phase']['frac'], d['roofline']['partition']['kernel_ms'], d.get('radix_overflow_rows'))
P
done
