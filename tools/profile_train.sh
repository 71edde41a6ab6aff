# the two training lines + the kernel stats of the OPQ one (part of tools/profile_round4.sh; run alone after a training change)
set -x
R=$PWD
O=$R/gpurun_out/r4
mkdir -p $O
python bench.py --workload train_opq --steps 25 --warmup 2 > $O/bench_train_opq.json 2> $O/bench_train_opq.err
python bench.py --workload train_pq --steps 25 --warmup 2 > $O/bench_train_pq.json 2> $O/bench_train_pq.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_train_opq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_opq -o s -- python $R/bench.py --workload train_opq --steps 25 --warmup 1 --no-cpu > /dev/null 2>&1
