#!/bin/bash
# round 5, call r: every aggregate / key-record test after the dictionary route went in
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_agg_string_gpu.py tests/test_agg_gpu.py tests/test_agg_packed_gpu.py tests/test_agg_keydict_gpu.py tests/test_join_keyrec_gpu.py tests/test_join_varlen_gpu.py tests/test_streamagg_gpu.py tests/test_pipeline_gpu.py tests/test_host_cpp_gpu.py tests/test_concurrency_gpu.py -m gpu -q 2>&1 | tail -8
