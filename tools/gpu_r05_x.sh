#!/bin/bash
# round 5, call x: Q3 with Sort + StreamAgg in place of the hash aggregate
mkdir -p gpurun_out
timeout 400 python3 tools/q3.py 100 --device-gen --verify --sort-agg 2>&1 | tail -3 | cut -c1-1200
timeout 400 python3 tools/q3.py 100 --device-gen 2>&1 | tail -1 | cut -c1-600
