#!/usr/bin/env python3
"""Per-kernel averages of every counter in the rocprofv3 CSV outputs under a directory (pmc_*/**/p_counter_collection.csv).
usage: python tools/pmc_csv_summary.py gpurun_out/r2"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
print("| pass | kernel | counter | launches | avg per launch |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rq::" not in k:
            continue
        a = acc[(k.replace("void ", "")[:64], row["Counter_Name"])]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    tag = f[len(root) + 1:].split("/")[0]
    for (k, c), (n, v) in sorted(acc.items()):
        print("| %s | %s | %s | %d | %.6g |" % (tag, k, c, n, v / n))
