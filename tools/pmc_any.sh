# PMC passes (one counter group per pass) of any command; prints per-kernel averages for kernels matching $PAT.
#   PAT=encode bash tools/pmc_any.sh <tag> <command...>
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p2 -o p -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p3 -o p -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/p4 -o p -- "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections, os
pat = os.environ.get("PAT", "rq::")
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k: continue
        a = acc[(k[:60], row["Counter_Name"])]
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items()):
        print("%-4s %-62s %-26s launches=%d avg=%.6g" % (f.split("/")[-2], k, c, n, v / n))
for f in sorted(glob.glob("$OUT/p1/*kernel_trace.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        if pat in k: print("trace", k, "n=%d avg_us=%.1f min_us=%.1f" % (len(v), sum(v) / len(v), min(v)))
PY
