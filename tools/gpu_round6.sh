#!/bin/bash
# short GPU check of the rowcodec slice (rank 4) + the contract tests added last: parity tests, then a small bench run
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 70 python -m pytest tests/test_storage_rowcodec_gpu.py -q -x --deselect tests/test_storage_rowcodec_gpu.py::test_full_size_round_trip_property > $O/rowcodec_tests.txt 2>&1; echo "pytest exit $?" >> $O/rowcodec_tests.txt
tail -5 $O/rowcodec_tests.txt
timeout 40 python -m pytest tests/test_sort_gpu.py::test_sort_call_sequence_and_cancel_contract tests/test_decode_gpu.py::test_decode_argument_contract tests/test_join_ordered_gpu.py::test_ordered_with_heavily_duplicated_keys -q > $O/late_tests.txt 2>&1; echo "pytest exit $?" >> $O/late_tests.txt
tail -3 $O/late_tests.txt
timeout 50 python tools/bench_rowcodec.py 4e6 > $O/rowcodec_bench.json 2> $O/rowcodec_bench.err; echo "bench exit $?"; cut -c1-600 $O/rowcodec_bench.json; tail -2 $O/rowcodec_bench.err
