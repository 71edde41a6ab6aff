# quick look at the training phases on the GPU box: bash tools/train_quick.sh
python bench.py --workload train_opq --steps 25 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train_opq', j['ms_per_step'], j['per_iteration_ms'], j.get('polar_factor'))"
python bench.py --workload train_pq --steps 25 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train_pq', j['ms_per_step'], j['per_iteration_ms'])"
