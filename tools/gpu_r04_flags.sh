#!/bin/bash
# round 4: NULL bitmap / selection flags on the fast path of k_da_partition2 (FLAGS instantiations)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_join_packed_gpu.py -m gpu -q -k "hot or null or NULL or selected or sel" 2>&1 | tail -4
python bench.py --no-cpu-baseline --steps 5 --only-extras variants_8d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['kernel_ms'])
for k,v in d['variants_8d'].items(): print(k, v.get('ms_per_probe_pass'), v.get('verified'), v.get('partition_kernel_ms'))"
