# Round-6 evidence: bench lines, rocprofv3 kernel stats and PMC passes of the same commands.  Run on the GPU box:
#   bash tools/profile_round6.sh [quick]      (outputs under gpurun_out/r6; tools/collect_profiles.py copies the summaries
#                                              to profiles/ and REWRITES profiles/r6_traffic.json from the PMC passes, so
#                                              the bytes bench.py replays cannot go stale after a kernel change)
set -x
R=$PWD
O=$R/gpurun_out/r6
mkdir -p $O
python -c "import rayuela_jl_amd as rq; from rayuela_jl_amd import _lib; print(_lib.lib().rq_version().decode().split('build ')[-1])" > $O/build_id.txt 2>/dev/null
python bench.py > $O/bench_pq.json 2> $O/bench_pq.err
python bench.py --workload opq > $O/bench_opq.json 2> $O/bench_opq.err
python bench.py --workload deep > $O/bench_deep.json 2> $O/bench_deep.err
python bench.py --k 10000 --no-cpu > $O/bench_pq_k10000.json 2> $O/bench_pq_k10000.err
python bench.py --workload sift1b --steps 3 --warmup 1 > $O/bench_sift1b_1gpu.json 2> $O/bench_sift1b_1gpu.err
# the exact per-GPU work of BASELINE config 5 on 8 GPUs: a 1.25e8-row shard, 1024 queries, k = 100
python bench.py --workload sift1b --rows 125000000 --steps 5 --warmup 1 --no-cpu > $O/bench_sift1b_shard.json 2> $O/bench_sift1b_shard.err
python bench.py --workload sift1b --inproc --gpus 1 --steps 3 --warmup 1 --no-cpu > $O/bench_sift1b_inproc.json 2> $O/bench_sift1b_inproc.err
RQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --rows 250000000 --steps 2 --warmup 1 --no-cpu > $O/bench_sift1b_2ranks_gloo.json 2> $O/bench_sift1b_2ranks_gloo.err
python tools/index_overhead.py > $O/index_overhead.md 2> $O/index_overhead.err
python bench.py --workload train_opq --steps 25 --warmup 2 > $O/bench_train_opq.json 2> $O/bench_train_opq.err
python bench.py --workload train_pq --steps 25 --warmup 2 > $O/bench_train_pq.json 2> $O/bench_train_pq.err
cd /tmp && export TMPDIR=/tmp
for w in pq opq deep; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python $R/bench.py --workload $w --no-cpu --no-host --no-ref1 --no-ab > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_opq -o s -- python $R/bench.py --workload train_opq --steps 25 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k10000 -o s -- python $R/bench.py --k 10000 --no-cpu --no-host --no-ref1 --no-ab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sift1b -o s -- python $R/bench.py --workload sift1b --steps 3 --warmup 1 --no-cpu --no-ab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sift1b_shard -o s -- python $R/bench.py --workload sift1b --rows 125000000 --steps 5 --warmup 1 --no-cpu --no-ab > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_deep_$c -o p -- python $R/bench.py --workload deep --no-cpu --no-host --no-ab --steps 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_sift1b_$c -o p -- python $R/bench.py --workload sift1b --no-cpu --no-ab --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_shard_$c -o p -- python $R/bench.py --workload sift1b --rows 125000000 --no-cpu --no-ab --steps 2 --warmup 1 > /dev/null 2>&1
done
# k = 10000 (the reference's default, src/Linscan.jl:10) and the Deep1M shape (m = 16): fabric traffic + LDS / VALU counters
# (VERDICT r5 Missing #4: no pass existed for adc_scan_kernel<8,false,true,true> nor an SQ_LDS pass for <16,...>)
K10="--k 10000 --no-cpu --no-host --no-ref1 --no-ab --steps 3"
DEEP="--workload deep --no-cpu --no-host --no-ref1 --no-ab --steps 3"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_k10000_$c -o p -- python $R/bench.py $K10 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_k10000_sq1 -o p -- python $R/bench.py $K10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_k10000_sq2 -o p -- python $R/bench.py $K10 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_k10000_ta -o p -- python $R/bench.py $K10 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_deep_sq1 -o p -- python $R/bench.py $DEEP > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_deep_sq2 -o p -- python $R/bench.py $DEEP > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_deep_ta -o p -- python $R/bench.py $DEEP > /dev/null 2>&1
# phase clocks of the scan kernel (tuning SCAN_STATS = 1: summed shader clocks of thread 0 per phase), un-profiled runs
cd $R
for t in "pq:" "k10000:--k 10000" "deep:--workload deep"; do
  RQ_SCAN_STATS=1 python bench.py ${t#*:} --no-cpu --no-host --no-ref1 --no-ab --steps 3 2> $O/phase_${t%%:*}.err > /dev/null
done
python tools/phase_clock.py $O > $O/phase_clock.md
( echo "# the balanced row order against the plain sort (tools/greedy_order_ab.py): pass model, ordering time, scan on the prepared base and inside a 1e4-query call"; echo; echo '```'; python tools/greedy_order_ab.py 8 2>/dev/null | grep 'm='; python tools/greedy_order_ab.py 16 2>/dev/null | grep 'm=\|^ '; echo '```' ) > $O/greedy_order.md
( echo "# m = 4: the integer pre-filter on / off (tools/m4_filter_ab.py; random codes, Gaussian tables; resident; HIP events)"; echo; echo '```'; python tools/m4_filter_ab.py 2>/dev/null; echo '```' ) > $O/m4_filter.md
cd /tmp
if [ "$1" != "quick" ]; then
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -o p -- python $R/bench.py --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_ta -o p -- python $R/bench.py --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_enc -o p -- python $R/bench.py --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_enc_deep -o p -- python $R/bench.py --workload deep --no-cpu --no-host --no-ref1 --no-ab --steps 3 > /dev/null 2>&1
fi
cd $R
python tools/pmc_csv_summary.py $O > $O/pmc_summary.txt
cat $O/pmc_summary.txt
du -sh $O
