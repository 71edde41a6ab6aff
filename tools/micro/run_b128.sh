# Round 6: 8 queries per ds_read_b64 against 16 queries per ds_read_b128 in the filter-only micro-kernel, on rows in arrival order
# and ordered by 15 key bits at 3 / 4 bits per byte; LDS and VALU counters of the ordered runs.   (GPU box) bash tools/micro/run_b128.sh
O=$PWD/gpurun_out/b128; mkdir -p $O
B=$PWD/tools/bin/filter_lean
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/filter_lean.hip -o $B
for thr in 62 48; do
( echo "## THR=$thr arrival order"; THR=$thr MICRO_B128_ONLY=1 $B; echo "## THR=$thr rows ordered by 15 bits, 3 bits per byte (b64 windows)"; THR=$thr ORDER=15 MICRO_B128_ONLY=1 $B; echo "## THR=$thr rows ordered by 15 bits, 4 bits per byte (b128 windows)"; THR=$thr ORDER=15 ORDER_BITS_PER_BYTE=4 MICRO_B128_ONLY=1 $B ) > $O/times_thr$thr.txt 2>&1
cat $O/times_thr$thr.txt
done
cd /tmp && export TMPDIR=/tmp
for cfg in "o3:ORDER=15" "o4:ORDER=15 ORDER_BITS_PER_BYTE=4"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs THR=48 MICRO_B128_ONLY=1 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2>&1
done
cd - > /dev/null
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for tag in ("o3", "o4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (O, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if "filt" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].replace("void ", "")[:28], r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("order by %s bits per byte" % tag[1], k, {c: "%.4g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
find $O -name "*.csv" -size +256k -delete
