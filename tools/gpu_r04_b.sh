#!/bin/bash
# round 4, GPU call B: the hot-key cases one process each (a fault in one must not hide the others); then the rest of call A's test list
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
: > $O/r04b_hot.txt
for shape in one_full_tile partial_last_tile null_bitmap sticky_overflow_flag; do
 for span in u16_entries u32_byte_cells u32_bit_cells; do
  for v in one_wg_per_cu two_wg_per_cu; do
    id="tests/test_join_packed_gpu.py::test_packed_hot_probe_key_both_partition_kernels[$shape-$span-$v]"
    timeout 120 python -m pytest "$id" -m gpu -q -x 2>&1 | grep -E "passed|failed|assert|Error|error|Abort|fault" | head -5 | sed "s|^|[$shape-$span-$v] |" >> $O/r04b_hot.txt
  done
 done
done
cat $O/r04b_hot.txt
(timeout 1500 python -m pytest tests/test_join_packed_gpu.py tests/test_comm_gpu.py tests/test_join_radix_gpu.py tests/test_agg_packed_gpu.py tests/test_agg_string_gpu.py \
   tests/test_storage_rowcodec_gpu.py tests/test_agg_gpu.py tests/test_expr_string_gpu.py -m gpu -q --maxfail=30 -k "not hot_probe_key" > $O/r04b_pytest.txt 2>&1; echo "rc=$?" >> $O/r04b_pytest.txt)
tail -30 $O/r04b_pytest.txt
echo done
