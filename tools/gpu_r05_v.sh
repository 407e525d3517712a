#!/bin/bash
# round 5, call v: golden vectors on the key-record and dictionary routes
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_join_keyrec_gpu.py tests/test_agg_keydict_gpu.py -m gpu -x -q 2>&1 | tail -25
