#!/bin/bash
# round 2: index scans (tsq_indexkeys_decode), memcomparable strings in tsq_rows_decode_chunks, the chunk wire format, the C++ host mirror
out=gpurun_out/${1:-r2i}
mkdir -p $out
timeout 900 python -m pytest tests/test_storage_indexscan_gpu.py tests/test_decode_chunks_gpu.py tests/test_chunk_wire_gpu.py tests/test_storage_coprocessor_gpu.py -x -q -m gpu -s > $out/pytest.txt 2>&1
tail -12 $out/pytest.txt
grep "tsq_indexkeys_decode:" $out/pytest.txt
(cd tinysql_amd/host && timeout 300 ./tsq_host_test > ../../$out/host.txt 2>&1; tail -3 ../../$out/host.txt)
