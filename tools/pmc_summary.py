#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel average of every collected counter and
the kernel durations.  usage: python tools/pmc_summary.py gpurun_out/pmc_*/p_results.db > profiles/...md"""
import sqlite3
import sys

print("| run | kernel | counter | launches | avg per launch |")
print("|---|---|---|---|---|")
for f in sys.argv[1:]:
    c = sqlite3.connect(f)
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, counter_name")
    for k, cn, cnt, avg in c.execute(q):
        if k.startswith("void rq::"):
            print("| %s | %s | %s | %d | %.6g |" % (f.split("/")[-2], k[5:60], cn, cnt, avg))
