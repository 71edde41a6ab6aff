# A/B of row orders on the scan kernel + the LDS conflict counters.  usage: bash tools/order_ab.sh <tag>
TAG=${1:-r4/order_ab}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python tools/order_ab.py --check > $OUT/sift.txt 2>&1
RQ_SCAN_STATS=1 python tools/order_ab.py --ks 1000,10000 --iters 2 > $OUT/sift_stats.txt 2>&1
python tools/order_ab.py --uniform --ks 1,1000 > $OUT/uniform.txt 2>&1
python tools/order_ab.py --deep --ks 1,1000 --orders arrival,b15 --check > $OUT/deep.txt 2>&1
python tools/order_ab.py --n 20000000 --nq 2048 --ks 100 --orders arrival,b15 > $OUT/n2e7.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for o in arrival b15; do
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_$o -o p -- python $R/tools/order_ab.py --ks 1000 --iters 2 --orders $o > /dev/null 2>&1
done
cd $R
python - <<PY > $OUT/pmc.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "adc_scan" not in k: continue
        a = acc[(k[:60], row["Counter_Name"])]
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items()):
        print("%-12s %-62s %-24s launches=%d avg=%.6g" % (f.split("/")[-2], k, c, n, v / n))
PY
cat $OUT/sift.txt $OUT/sift_stats.txt $OUT/uniform.txt $OUT/deep.txt $OUT/n2e7.txt $OUT/pmc.txt
