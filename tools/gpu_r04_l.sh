#!/bin/bash
# round 4, run L: the default bench line at the final commit (traffic file corrected), string expressions after the one-call VecEvalString
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_expr_string_gpu.py tests/test_shared_plan_cpu.py -q 2>&1 | tail -6 > $O/r04l_pytest.txt
SECONDS=0
timeout 600 python bench.py > $O/r04l_bench.json 2> $O/r04l_bench.err
echo "default bench.py wall seconds: $SECONDS" > $O/r04l_bench.time
ls -la $O | grep r04l
echo done
