# Round 6: chunk pacing of the big-base scan (SCAN_PACE) -- time and fabric traffic per launch on the 1.25e8-row shard of config 5
# (and optionally 1e9 rows).  usage (GPU box): bash tools/pace_sweep.sh [rows]   -> gpurun_out/pace/summary.md
R=$PWD; O=$R/gpurun_out/pace; mkdir -p $O
ROWS=${1:-125000000}
B="--workload sift1b --rows $ROWS --no-cpu --no-ab --no-ref1 --steps 3 --warmup 1"
echo "| setting | ms per step (events, un-profiled) | FETCH_SIZE x 2 (GB / launch) | WRITE_SIZE (GB / launch) | x code bytes |" > $O/summary.md
echo "|---|---|---|---|---|" >> $O/summary.md
for cfg in "off:RQ_SCAN_PACE=0" "lag0:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=0" "lag1:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=1" "lag2:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=2" "lag4:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=4" "lag8:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=8" "lag2v4:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=2 RQ_SCAN_PACE_VOTES=4" "lag2_slack0:RQ_SCAN_PACE=1 RQ_SCAN_PACE_LAG=2 RQ_SCAN_XCD_SLACK=0" $PACE_EXTRA; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py $B > $O/bench_$tag.json 2> $O/bench_$tag.err
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && env $envs rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${tag}_$c -o p -- python $R/bench.py $B > /dev/null 2>&1 )
  done
  python - $O $tag $ROWS >> $O/summary.md <<'PY'
import csv, glob, json, sys
O, tag, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = json.loads(open("%s/bench_%s.json" % (O, tag)).read().strip().splitlines()[-1])
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (O, tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "adc_scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
    val[c] = tot / max(n, 1) * 1024.0            # KiB -> bytes per launch
f2 = 2.0 * val["FETCH_SIZE"]
print("| %s | %.3f (kernel %.3f) | %.2f | %.2f | %.1f |" % (tag, d["ms_per_step"], d["roofline"]["kernel_ms"], f2 / 1e9, val["WRITE_SIZE"] / 1e9, (f2 + val["WRITE_SIZE"]) / (rows * 8.0)))
PY
done
find $O -name "*.csv" -size +256k -delete
cat $O/summary.md
