# PMC passes of the scan kernel on an ORDERED base (tools/order_perf.py --modes ob).  usage: bash tools/pmc_order.sh <tag> [order_perf args]
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
ARGS="${@:---modes ob --ks 1000 --iters 3}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python $R/tools/order_perf.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python $R/tools/order_perf.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/ta -o p -- python $R/tools/order_perf.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/g -o p -- python $R/tools/order_perf.py $ARGS > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "adc_scan" not in k: continue
        a = acc[(k[:60], row["Counter_Name"])]
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items()):
        print("%-6s %-50s %-24s launches=%d avg=%.6g" % (f.split("/")[-2], k[:50], c, n, v / n))
PY
