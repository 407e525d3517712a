#!/bin/bash
# round 5, run E: Q3 SF 100 with the result pulled into pinned memory + verified per group against numpy; the new tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_streamagg_gpu.py tests/test_decode_gpu.py tests/test_agg_packed_gpu.py -m gpu -q -k "stream or single_stream or hot_keys or argument_contract or errors_are" > $O/r05_e_tests.txt 2>&1
tail -8 $O/r05_e_tests.txt
timeout 600 python tools/q3.py 100 --device-gen --verify > $O/r05_e_q3.json 2> $O/r05_e_q3.err
tail -3 $O/r05_e_q3.err
python3 -c "
import json
d=json.loads(open('$O/r05_e_q3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('groups','best_s','exec_s_result_in_hbm','verified_against_numpy','result_checksum')})"
