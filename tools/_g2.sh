python -m pytest tests/test_gpu_nonfinite.py tests/test_gpu_order.py -x -q 2>&1 | tail -30
R=$PWD; O=$R/gpurun_out/x2; mkdir -p $O
K10="--k 10000 --no-cpu --no-host --no-ref1 --no-ab --steps 3"
DEEP="--workload deep --no-cpu --no-host --no-ref1 --no-ab --steps 3"
for t in "pq:" "k10000:--k 10000" "deep:--workload deep"; do
  RQ_SCAN_STATS=1 python bench.py ${t#*:} --no-cpu --no-host --no-ref1 --no-ab --steps 3 2> $O/phase_${t%%:*}.err > $O/bench_${t%%:*}.json
done
python tools/phase_clock.py $O > $O/phase_clock.md; cat $O/phase_clock.md
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_k10000_sq1 -o p -- python $R/bench.py $K10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_k10000_sq2 -o p -- python $R/bench.py $K10 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_deep_sq1 -o p -- python $R/bench.py $DEEP > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_deep_sq2 -o p -- python $R/bench.py $DEEP > /dev/null 2>&1
cd $R; python tools/pmc_csv_summary.py $O | grep -v "order_\|encode\|rotate\|widen\|merge" > $O/pmc.md; cat $O/pmc.md
find $O -name "*.csv" -size +1M -delete
