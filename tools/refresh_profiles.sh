#!/bin/bash
# Development-container driver: profile round on a GPU box (gpurun), collect into profiles/, then re-run the four headline
# bench lines WITH the fresh replay files in place (they are bound to the library build) and store those lines too.
#   bash tools/refresh_profiles.sh
set -e
cd "$(dirname "$0")/.."
rm -rf gpurun_out/r5 gpurun_out/r5b
/usr/local/graft/bin/gpurun --timeout 2400 -- 'rm -rf gpurun_out/r5; bash tools/profile_round5.sh > gpurun_out/r5_profile.log 2>&1; tail -2 gpurun_out/r5_profile.log'
python tools/collect_profiles.py r5 | head -12
/usr/local/graft/bin/gpurun --timeout 1200 -- 'mkdir -p gpurun_out/r5b; python bench.py > gpurun_out/r5b/bench_pq.json 2>/dev/null; python bench.py --workload opq > gpurun_out/r5b/bench_opq.json 2>/dev/null; python bench.py --workload deep > gpurun_out/r5b/bench_deep.json 2>/dev/null; python bench.py --k 10000 --no-cpu > gpurun_out/r5b/bench_pq_k10000.json 2>/dev/null; echo done'
python - <<'PY'
import json
for w in ("pq", "opq", "deep", "pq_k10000"):
    d = json.loads(open("gpurun_out/r5b/bench_%s.json" % w).read().strip().splitlines()[-1])
    r = d["roofline"]; e = d.get("encode") or {}; er = e.get("roofline") or {}
    print(w, d["ms_per_step"], "kernel", r.get("kernel_ms"), "frac", r["frac"], "traffic", r["traffic"], "| enc", e.get("ms_per_step"), er.get("bound"), er.get("frac"))
    open("profiles/r5_bench_%s.json" % w, "w").write(json.dumps(d) + "\n")
PY
