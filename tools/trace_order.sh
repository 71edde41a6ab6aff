R=$PWD
OUT=$R/gpurun_out/r4/trace_raw1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/tools/order_perf.py --ks 100,1000 --modes raw1,ob --iters 10 > $OUT/out.txt 2>&1
cd $R
cat $OUT/out.txt | grep -v amdgpu
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    acc[row["Kernel_Name"][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "rq::" in k: print("%-72s n=%4d avg_us=%9.1f min=%9.1f max=%9.1f" % (k, len(v), sum(v)/len(v), min(v), max(v)))
PY
