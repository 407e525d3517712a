#!/bin/bash
# round 5, call o: outer-side filters + outer-join conditions over duplicate build keys on the packed route; memcomparable string encode
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_storage_rows_encode_gpu.py tests/test_join_gpu.py tests/test_pipeline_gpu.py tests/test_host_cpp_gpu.py -m gpu -q 2>&1 | tail -15
