R=$PWD
OUT=$R/gpurun_out/r4/trace_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --workload ${1:-train_opq} --steps 25 --warmup 1 --no-cpu > $OUT/out.txt 2>&1
cd $R
python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    acc[row["Kernel_Name"][:90]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-92s n=%4d avg_us=%9.1f min=%9.1f max=%9.1f total_ms=%8.2f" % (k, len(v), sum(v)/len(v), min(v), max(v), sum(v)/1e3))
PY
