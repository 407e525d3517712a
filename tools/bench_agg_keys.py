#!/usr/bin/env python3
"""GROUP BY over several integer key columns alone (bench.py's `agg_two_keys_*` side measurements), for rocprofv3:
   python tools/bench_agg_keys.py [ma mb [rows]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402

ma, mb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 100)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 250_000_000
ctx = _lib.Context(0)
ctx.reserve(32 << 30)
print(json.dumps(bench.extra_two_keys(ctx, abi, _lib, n=n, ma=ma, mb=mb)))
ctx.close()
