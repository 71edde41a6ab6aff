#!/usr/bin/env python3
"""Phase clock of the scan kernel from `RQ_SCAN_STATS=1 python bench.py ...` runs (stderr lines "scan_stats {...}" written by
bench.py; the counters are summed shader clocks of thread 0 of every workgroup, rq_scan_stats / rq_scan_finish_stats).
usage: python tools/phase_clock.py gpurun_out/r6   (reads phase_<tag>.err)"""
import glob
import json
import os
import sys

root = sys.argv[1]
print("| run | lut | sample | stream | in-stream cuts | final cut | finish | of the finish: p9 / p10 / p11 | items | exact fallbacks | first-block alive rows | bucket finish: items / look skips / select+sort |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(root, "phase_*.err"))):
    st = None
    for line in open(f):
        if line.startswith("scan_stats "):
            st = json.loads(line[len("scan_stats "):])
    if st is None:
        continue
    tot = sum(st[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
    pc = lambda k: "%.1f %%" % (100.0 * st[k] / tot)       # noqa: E731
    print("| %s | %s | %s | %s | %s | %s | %s | %s / %s / %s | %d | %d | %.2f %% | %d / %d / %d |" % (
        os.path.basename(f)[6:-4], pc("lut"), pc("sample"), pc("stream"), pc("cuts"), pc("final_cut"), pc("sort_write"),
        pc("sort_load"), pc("sort_stages"), pc("sort_out"), st["n_items"], st["n_fallbacks"],
        100.0 * st["first_block_pushed"] / max(1, st["first_block_rows"]),
        st.get("bf_items", 0), st.get("bf_look_skips", 0), st.get("bf_select_sort", 0)))
print()
print("p9 / p10 / p11: K <= 1024 bucket finish = range + histogram / scan + scatter / rank + write; K > 1024 sample sort with the "
      "distance map = range + mean / histogram / scan + scatter (the ranking is the rest of `finish`); select + sort fallback = "
      "load / sort stages / write-out.  Shares of the summed per-phase clocks; `stream` includes the exact re-evaluation of alive rows.")
