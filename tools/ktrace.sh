# Per-kernel times of a command under rocprofv3 --kernel-trace.  usage: bash tools/ktrace.sh <tag> <command...>
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o p -- "$@" > $OUT/cmd.log 2>&1 )
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].replace("void ", "")[:90]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if "rq::" in k: print("%-92s n=%-4d avg_us=%9.1f min_us=%9.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
