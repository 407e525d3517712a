#!/bin/bash
# round 4: NULL bitmap / selection flags on the fast path of k_da_partition_cols
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_join_packed_gpu.py -m gpu -q -k "travelling or selected" 2>&1 | tail -4
python bench.py --no-cpu-baseline --steps 3 --only-extras materialising,materialising_nullable_left_outer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('materialising','materialising_nullable_left_outer'):
    v=d[k]; print(k, {x:v.get(x) for x in ('ms','repeated_probe_pass_ms','packed_prepare_ms','verified')})"
