#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the side kernels: decode, sort, build.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c -d $O/pmc_dec_$c -o d --output-format csv -- python $R/tools/bench_decode.py 10000000 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc $c -d $O/pmc_sort_$c -o s --output-format csv -- python $R/tools/bench_sort.py 30000000 > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc $c -d $O/pmc_build_$c -o b --output-format csv -- python $R/tools/bench_build.py 30000000 > /dev/null 2>&1
done
python $R/tools/summarize_prof.py $O/side_pmc_summary.txt --pmc $O/pmc_dec_FETCH_SIZE/d_counter_collection.csv --pmc $O/pmc_dec_WRITE_SIZE/d_counter_collection.csv \
  --pmc $O/pmc_sort_FETCH_SIZE/s_counter_collection.csv --pmc $O/pmc_sort_WRITE_SIZE/s_counter_collection.csv \
  --pmc $O/pmc_build_FETCH_SIZE/b_counter_collection.csv --pmc $O/pmc_build_WRITE_SIZE/b_counter_collection.csv \
  --note "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (rocprofv3 --pmc, one counter per pass); decode 1e7 rows x 4 cols (264.7 MB in), sort 3e7 rows, build 3e7 rows"
grep -E "k_dec_|k_sort_scatter|k_sort_tilehist|k_build_images|k_radix_sub|k_radix_partition<1024, 8, 4, 0, true>|k_gather_rows" $O/side_pmc_summary.txt | cut -c1-160 | head -30
