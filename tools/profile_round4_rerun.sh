set -x
R=$PWD; O=$R/gpurun_out/r4
python bench.py --k 10000 --no-cpu > $O/bench_pq_k10000.json 2> $O/bench_pq_k10000.err
python bench.py --workload sift1b --steps 3 --warmup 1 > $O/bench_sift1b_1gpu.json 2> $O/bench_sift1b_1gpu.err
python bench.py --workload sift1b --rows 125000000 --steps 5 --warmup 1 --no-cpu > $O/bench_sift1b_shard.json 2> $O/bench_sift1b_shard.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_k10000 $O/stats_sift1b $O/stats_sift1b_shard $O/pmc_sift1b_* $O/pmc_shard_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k10000 -o s -- python $R/bench.py --k 10000 --no-cpu --no-host --no-ref1 --no-ab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sift1b -o s -- python $R/bench.py --workload sift1b --steps 3 --warmup 1 --no-cpu --no-ab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sift1b_shard -o s -- python $R/bench.py --workload sift1b --rows 125000000 --steps 5 --warmup 1 --no-cpu --no-ab > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_sift1b_$c -o p -- python $R/bench.py --workload sift1b --no-cpu --no-ab --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_shard_$c -o p -- python $R/bench.py --workload sift1b --rows 125000000 --no-cpu --no-ab --steps 2 --warmup 1 > /dev/null 2>&1
done
cd $R
python tools/pmc_csv_summary.py $O > $O/pmc_summary.txt
tail -c 300 $O/bench_sift1b_1gpu.err
