#!/bin/bash
# The ONE runner for work on the GPU box (replaces the per-experiment gpu_r0N_x.sh scripts of rounds 4-5).  Run through gpurun from
# the repo root, several steps per lease:
#   gpurun --timeout 900 -- 'tools/gpu.sh tests; tools/gpu.sh prof r06_mat -- python bench.py --no-cpu-baseline --only-extras materialising'
# Steps (everything lands under gpurun_out/, summaries are ready to be copied into profiles/):
#   tests [pytest args]        python -m pytest tests -m gpu -q  -> <tag>_pytest_gpu.txt   (tag = $TAG or r06)
#   bench [bench args]         the driver's command              -> <tag>_bench.json (+ bench_extras.json -> <tag>_bench_extras.json)
#   run  NAME -- CMD...        CMD, stdout -> NAME.out, stderr -> NAME.err
#   prof NAME -- CMD...        rocprofv3 --kernel-trace --stats of CMD      -> NAME_rocprof.txt
#   pmc  NAME -- CMD...        two --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, one counter per pass as the guide prescribes)
#                              -> NAME_pmc.txt (one line per kernel and grid size)
#   sq   NAME -- CMD...        one --pmc pass of SQ counters (busy / wait split) -> NAME_sq.txt
#   all  NAME -- CMD...        prof + pmc of the same command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
TAG=${TAG:-r06}
mkdir -p "$O"
export TMPDIR=/tmp
step=$1
shift
name=""
if [ "$step" != tests ] && [ "$step" != bench ]; then
  name=$1
  shift
  [ "$1" = "--" ] && shift
fi
# the profiler runs from /tmp (rocprofv3 writes next to its cwd): arguments that name files of the repo become absolute paths
abs_args() {
  ARGS=()
  for a in "$@"; do
    if [ -e "$R/$a" ] && [ "${a#/}" = "$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi
  done
}
prof() {
  abs_args "$@"; set -- "${ARGS[@]}"
  ( cd /tmp && timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d "$O/.prof_$name" -o p --output-format csv -- "$@" > "$O/${name}_prof.out" 2> "$O/${name}_prof.err" )
  python3 "$R/tools/summarise_prof.py" stats "$(find "$O/.prof_$name" -name '*kernel_stats.csv' | head -1)" > "$O/${name}_rocprof.txt" 2>&1
  rm -rf "$O/.prof_$name"
  head -${HEAD:-14} "$O/${name}_rocprof.txt" | cut -c1-200
}
pmc() {
  abs_args "$@"; set -- "${ARGS[@]}"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc $c -d "$O/.pmc_${name}_$c" -o p --output-format csv -- "$@" > "$O/${name}_pmc_$c.out" 2> "$O/${name}_pmc_$c.err" )
  done
  python3 "$R/tools/summarise_prof.py" pmc-by-grid $(find "$O/.pmc_${name}_FETCH_SIZE" "$O/.pmc_${name}_WRITE_SIZE" -name '*counter_collection.csv') > "$O/${name}_pmc.txt" 2>&1
  rm -rf "$O"/.pmc_${name}_*
  grep -c . "$O/${name}_pmc.txt"
}
cd "$R"
case $step in
  tests)
    ( time timeout ${TEST_TIMEOUT:-1500} python3 -m pytest tests -m gpu -q -x "$@" ) > "$O/${TAG}_pytest_gpu.txt" 2>&1
    tail -6 "$O/${TAG}_pytest_gpu.txt" ;;
  bench)
    ( time timeout 900 python3 bench.py "$@" > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err" ) 2> "$O/${TAG}_bench.time"
    [ -f "$O/bench_extras.json" ] && cp "$O/bench_extras.json" "$O/${TAG}_bench_extras.json"
    cat "$O/${TAG}_bench.json"; tail -3 "$O/${TAG}_bench.time"; tail -3 "$O/${TAG}_bench.err" ;;
  run)
    ( time timeout ${RUN_TIMEOUT:-900} "$@" > "$O/$name.out" 2> "$O/$name.err" ) 2> "$O/$name.time"
    tail -${TAIL:-20} "$O/$name.out"; tail -5 "$O/$name.err"; tail -3 "$O/$name.time" ;;
  prof) prof "$@" ;;
  pmc) pmc "$@" ;;
  sq)
    abs_args "$@"; set -- "${ARGS[@]}"
    ( cd /tmp && timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
        -d "$O/.sq_$name" -o p --output-format csv -- "$@" > "$O/${name}_sq.out" 2> "$O/${name}_sq.err" )
    python3 "$R/tools/summarise_prof.py" pmc-by-grid $(find "$O/.sq_$name" -name '*counter_collection.csv') > "$O/${name}_sq.txt" 2>&1
    rm -rf "$O/.sq_$name"
    grep -c . "$O/${name}_sq.txt" ;;
  all) prof "$@"; pmc "$@" ;;
  *) echo "unknown step $step" >&2; exit 2 ;;
esac
