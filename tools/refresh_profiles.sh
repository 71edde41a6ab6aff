#!/bin/bash
# Development-container driver: profile round on a GPU box (gpurun), collect into profiles/, then re-run the six bench lines that replay traffic WITH the fresh replay files in place (they are bound to the library build) and store those lines too.
#   bash tools/refresh_profiles.sh
set -e
cd "$(dirname "$0")/.."
rm -rf gpurun_out/r6 gpurun_out/r6b
/usr/local/graft/bin/gpurun --timeout 2400 -- 'rm -rf gpurun_out/r6; bash tools/profile_round6.sh > gpurun_out/r6_profile.log 2>&1; tail -2 gpurun_out/r6_profile.log'
python tools/collect_profiles.py r6 | head -12
/usr/local/graft/bin/gpurun --timeout 1200 -- 'mkdir -p gpurun_out/r6b; python bench.py > gpurun_out/r6b/bench_pq.json 2>/dev/null; python bench.py --workload opq > gpurun_out/r6b/bench_opq.json 2>/dev/null; python bench.py --workload deep > gpurun_out/r6b/bench_deep.json 2>/dev/null; python bench.py --k 10000 --no-cpu > gpurun_out/r6b/bench_pq_k10000.json 2>/dev/null; python bench.py --workload sift1b --steps 3 --warmup 1 > gpurun_out/r6b/bench_sift1b_1gpu.json 2>/dev/null; python bench.py --workload sift1b --rows 125000000 --steps 5 --warmup 1 --no-cpu > gpurun_out/r6b/bench_sift1b_shard.json 2>/dev/null; echo done'
python - <<'PY'
import json
for w in ("pq", "opq", "deep", "pq_k10000", "sift1b_1gpu", "sift1b_shard"):
    d = json.loads(open("gpurun_out/r6b/bench_%s.json" % w).read().strip().splitlines()[-1])
    r = d["roofline"]; e = d.get("encode") or {}; er = e.get("roofline") or {}
    print(w, d["ms_per_step"], "kernel", r.get("kernel_ms"), "frac", r["frac"], "traffic", r["traffic"], "| enc", e.get("ms_per_step"), er.get("bound"), er.get("frac"))
    open("profiles/r6_bench_%s.json" % w, "w").write(json.dumps(d) + "\n")
PY
