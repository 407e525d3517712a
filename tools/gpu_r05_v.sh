#!/bin/bash
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_agg_keydict_gpu.py tests/test_agg_string_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | tail -4
