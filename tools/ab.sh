# A/B of library builds under ab_libs/: usage  bash tools/ab.sh "base thr95 thr127" [order_perf args]
LIBS=$1; shift
for l in $LIBS; do
  echo "== $l"
  RAYUELA_HIP_LIB=$PWD/ab_libs/$l.so python tools/order_perf.py --modes ob "$@" 2>&1 | grep -v "amdgpu.ids\|order_rows"
  RAYUELA_HIP_LIB=$PWD/ab_libs/$l.so RQ_SCAN_STATS=1 python tools/order_perf.py --modes ob --ks 100,1000 --iters 2 2>&1 | grep -o "K=[0-9]* .*ms\|first_block_alive=.*"
done
