cd $GRAFT_REPO_ROOT
for k in "DAAGG_LOG2C=11" "DAAGG_LOG2C=11 --knob DAAGG_PART2=0" "DAAGG_LOG2C=10"; do
  python bench.py --no-cpu-baseline --steps 2 --only-extras c3_agg_1e9_1e6 --knob $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['c3_agg_1e9_1e6']; print(d.get('knobs'), v.get('ms'), v.get('verified'), v.get('error'))"
done
