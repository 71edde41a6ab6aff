# FETCH_SIZE / WRITE_SIZE of the scan kernel for `bench.py --workload sift1b --rows $1` (extra env passes through).  Usage on the GPU box:
#   bash tools/fetch_probe.sh ROWS TAG
R=$PWD; O=$R/gpurun_out/r3/probe_$2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --workload sift1b --rows $1 --no-cpu --no-ref1 --steps 2 --warmup 1 > $O/bench_$c.json 2>/dev/null
done
cd $R
python tools/pmc_csv_summary.py $O | grep adc_scan | sed "s/^/$2 rows=$1 /"
