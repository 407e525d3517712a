#!/bin/bash
# round 2: the chunk wire format — GPU parity + timing
out=gpurun_out/${1:-r2w}
mkdir -p $out
timeout 600 python -m pytest tests/test_chunk_wire_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -15 $out/pytest.txt
timeout 300 python tools/bench_wire.py 5e7 > $out/wire.json 2> $out/wire.err
cat $out/wire.json; tail -3 $out/wire.err
