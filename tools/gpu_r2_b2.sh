#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2b2}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/bench_decode.py 2e6 --chunks > $O/decode_chunks.json 2> $O/decode_chunks.err; cat $O/decode_chunks.json; tail -3 $O/decode_chunks.err
