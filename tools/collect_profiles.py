#!/usr/bin/env python3
"""Copy the judged summaries of gpurun_out/<round> (written by tools/profile_round<N>.sh on the GPU box) into profiles/:
bench lines, the library's rows of the rocprofv3 kernel stats, the PMC table and the FETCH/WRITE traffic per launch
(profiles/<round>_traffic.json is REWRITTEN from the PMC passes every time, so the bytes bench.py replays follow the kernels).
usage: python tools/collect_profiles.py [r3]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else "r6"
O = "gpurun_out/" + RND
for f in ("bench_pq", "bench_opq", "bench_deep", "bench_pq_k10000", "bench_sift1b_1gpu", "bench_sift1b_shard", "bench_sift1b_inproc", "bench_train_opq", "bench_train_pq", "bench_sift1b_2ranks_gloo"):
    if os.path.exists("%s/%s.json" % (O, f)):
        open("profiles/%s_%s.json" % (RND, f), "w").write(open("%s/%s.json" % (O, f)).read().strip().splitlines()[-1] + "\n")
for w in ("pq", "opq", "deep", "k10000", "sift1b", "sift1b_shard", "train_opq"):
    g = glob.glob("%s/stats_%s/**/s_kernel_stats.csv" % (O, w), recursive=True)
    if not g:
        continue
    rows = list(csv.reader(open(g[0])))
    csv.writer(open("profiles/%s_bench_%s_kernel_stats.csv" % (RND, w), "w")).writerows([rows[0]] + [r for r in rows[1:] if "rq::" in r[0]])
shutil.copy(O + "/pmc_summary.txt", "profiles/%s_pmc_counters.md" % RND)
for extra in ("phase_clock.md", "m4_filter.md", "shape_sweep.md", "greedy_order.md"):
    if os.path.exists(O + "/" + extra):
        shutil.copy(O + "/" + extra, "profiles/%s_%s" % (RND, extra))
if os.path.exists(O + "/index_overhead.md"):
    shutil.copy(O + "/index_overhead.md", "profiles/%s_index_overhead.md" % RND)
shape = {"pmc_FETCH_SIZE": "adc_scan_kernel<8> n=1000000 nq=10000 k=1000", "pmc_WRITE_SIZE": "adc_scan_kernel<8> n=1000000 nq=10000 k=1000",
         "pmc_k10000_FETCH_SIZE": "adc_scan_kernel<8> n=1000000 nq=10000 k=10000", "pmc_k10000_WRITE_SIZE": "adc_scan_kernel<8> n=1000000 nq=10000 k=10000",
         "pmc_deep_FETCH_SIZE": "adc_scan_kernel<16> n=1000000 nq=10000 k=1000", "pmc_deep_WRITE_SIZE": "adc_scan_kernel<16> n=1000000 nq=10000 k=1000",
         "pmc_sift1b_FETCH_SIZE": "adc_scan_kernel<8> n=1000000000 nq=1024 k=100", "pmc_sift1b_WRITE_SIZE": "adc_scan_kernel<8> n=1000000000 nq=1024 k=100",
         "pmc_shard_FETCH_SIZE": "adc_scan_kernel<8> n=125000000 nq=1024 k=100", "pmc_shard_WRITE_SIZE": "adc_scan_kernel<8> n=125000000 nq=1024 k=100"}
build = open(O + "/build_id.txt").read().strip() if os.path.exists(O + "/build_id.txt") else "?"
t = {}
enc = {}
for line in open("profiles/%s_pmc_counters.md" % RND):
    m = re.match(r"\| (pmc_\S+) \| (.*?) \| (\S+) \| (\d+) \| (\S+) \|", line)
    if not m:
        continue
    if m.group(3) in ("FETCH_SIZE", "WRITE_SIZE") and "adc_scan" in m.group(2) and m.group(1) in shape:
        # bound to what was measured: the kernel instantiation (as rq_last_scan_kernel() spells it) and the library build
        kern = re.sub(r"^rq::", "", m.group(2)).split("(rq::")[0]
        e = t.setdefault(shape[m.group(1)], {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py of that "
                                                       "workload (tools/profile_round%s.sh); profiles/%s_pmc_counters.md" % (RND[1:], RND),
                                             "kernel": kern, "build": build})
        e[m.group(3) + "_KiB"] = float(m.group(5))
    if m.group(1) in ("pmc_enc", "pmc_enc_deep") and m.group(3) in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA") and "encode_" in m.group(2):
        # instruction counters of the encode's three launches (tables + filter + exact pass), summed per encode call
        key = "encode_pq_filter_kernel sub=%d" % (16 if m.group(1) == "pmc_enc" else 6)
        e = enc.setdefault(key, {"rows": 1000000, "build": build, "SQ_INSTS_VALU": 0.0, "SQ_INSTS_MFMA": 0.0, "launches": [],
                                 "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA ... on bench.py (tools/profile_round%s.sh); "
                                           "profiles/%s_pmc_counters.md" % (RND[1:], RND)})
        e[m.group(3)] += float(m.group(5))
        if m.group(3) == "SQ_INSTS_VALU":
            e["launches"].append(re.sub(r"^rq::", "", m.group(2)).split("(rq::")[0])
json.dump(t, open("profiles/%s_traffic.json" % RND, "w"), indent=1)
json.dump(enc, open("profiles/%s_encode_counters.json" % RND, "w"), indent=1)
for f in sorted(os.listdir("profiles")):
    if f.startswith(RND + "_bench") and f.endswith(".json"):
        d = json.load(open("profiles/" + f))
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], "frac", r.get("frac"), "f32roof", (r.get("f32_table_roof") or {}).get("frac"),
              "enc", (d.get("encode") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              (d.get("cpu_baseline") or {}).get("gpu_matches_cpu_bit_exact"))
print(json.dumps(t, indent=1))
