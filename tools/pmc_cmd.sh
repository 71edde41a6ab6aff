# PMC passes (one counter group per pass, no trace domains besides --kernel-trace) of any command; per-kernel averages.
# usage: bash tools/pmc_cmd.sh <tag> <kernel-substring> <command...>     groups: $PMC_GROUPS (semicolon-separated) or the default set
TAG=$1; KSUB=$2; shift; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
GROUPS_DEF="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY;SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS;SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
IFS=';' read -ra GR <<< "${PMC_GROUPS:-$GROUPS_DEF}"
i=0
for g in "${GR[@]}"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/cmd$i.log 2>&1 )
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "$KSUB" not in k: continue
        a = acc[(k.replace("void ", "")[:60], row["Counter_Name"])]
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items()):
        print("%-62s %-26s launches=%d avg=%.6g" % (k, c, n, v / n))
PY
