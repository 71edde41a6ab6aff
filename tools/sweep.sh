# usage: bash tools/sweep.sh "ENV=VAL ..." ...   -- one bench.py scan timing per argument (K from $KS, default "100 1000 10000")
KS=${KS:-"100 1000 10000"}
for cfg in "$@"; do
  for k in $KS; do
    ms=$(env $cfg python bench.py --workload ${WL:-pq} --k $k --no-cpu --no-host --steps 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$cfg K=$k $ms"
  done
done
