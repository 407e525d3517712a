#!/bin/bash
# round 4, GPU call A: the shared-images distributed plan, both partition kernels under hot keys, the knob conversion, bench smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(timeout 1200 python -m pytest tests/test_join_packed_gpu.py tests/test_comm_gpu.py tests/test_join_radix_gpu.py tests/test_agg_packed_gpu.py tests/test_agg_string_gpu.py \
   tests/test_storage_rowcodec_gpu.py tests/test_agg_gpu.py -m gpu -q --maxfail=30 > $O/r04a_pytest.txt 2>&1; echo "rc=$?" >> $O/r04a_pytest.txt)
tail -5 $O/r04a_pytest.txt
B="python bench.py --no-cpu-baseline"
timeout 600 $B > $O/r04a_bench.json 2> $O/r04a_bench.err
timeout 300 $B --no-extras --force-dist > $O/r04a_dist_shared_w1.json 2> $O/r04a_dist.err
timeout 300 $B --no-extras --force-dist --emulate-world 2 > $O/r04a_dist_emu2.json 2>> $O/r04a_dist.err
timeout 300 $B --no-extras --force-dist --emulate-world 4 > $O/r04a_dist_emu4.json 2>> $O/r04a_dist.err
timeout 300 $B --no-extras --force-dist --emulate-world 8 > $O/r04a_dist_emu8.json 2>> $O/r04a_dist.err
timeout 300 $B --no-extras --force-dist --emulate-world 8 --knob DA_PARTITION=2 > $O/r04a_dist_emu8_part2.json 2>> $O/r04a_dist.err
timeout 300 $B --no-extras --force-dist --dist-plan exchange > $O/r04a_dist_exchange_w1.json 2>> $O/r04a_dist.err
timeout 300 $B --only-extras wide_keys_31bit_unique_bit_cells,two_key_columns_count --knob DA_PARTITION=2 > $O/r04a_wide_part2.json 2>> $O/r04a_dist.err
timeout 300 $B --only-extras wide_keys_31bit_unique_bit_cells,two_key_columns_count --knob DA_PARTITION=1 > $O/r04a_wide_part1.json 2>> $O/r04a_dist.err
tail -3 $O/r04a_dist.err
echo done
