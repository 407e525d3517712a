#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( for b in 10 11; do for j in 2 3 4 5 6; do for m in 80 81 82 83 84 85; do timeout 120 ./tools/probe_exp $b $m $j; done; done; done ) > $O/probe_exp3.txt 2>&1
export TMPDIR=/tmp
cd /tmp
cat $O/probe_exp3.txt
