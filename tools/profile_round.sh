set -x
mkdir -p gpurun_out/r1f
python bench.py > gpurun_out/r1f/bench_pq.json 2> gpurun_out/r1f/bench_pq.err
python bench.py --workload opq > gpurun_out/r1f/bench_opq.json 2> gpurun_out/r1f/bench_opq.err
python bench.py --workload deep > gpurun_out/r1f/bench_deep.json 2> gpurun_out/r1f/bench_deep.err
python bench.py --k 10000 --no-cpu > gpurun_out/r1f/bench_pq_k10000.json 2> gpurun_out/r1f/bench_pq_k10000.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/stats_pq -o s -- python $R/bench.py --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/stats_opq -o s -- python $R/bench.py --workload opq --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/stats_k10000 -o s -- python $R/bench.py --k 10000 --no-cpu > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r1f/pmc_$c -o p -- python $R/tools/perf.py scan encode --ks 1000 --iters 2 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r1f/pmc_sq -o p -- python $R/tools/perf.py scan --ks 1000,10000 --iters 2 > /dev/null 2>&1
cd $R
find gpurun_out/r1f -name "*.csv" | head -30
du -sh gpurun_out/r1f
