#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per (kernel, counter): mean per dispatch."""
import csv
import collections
import glob
import re
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            m = re.search(r"(klt_kernel|pyr_level_kernel<\w+>|ekf_\w+)", name)
            k = (m.group(1) if m else name[:48].replace(",", ";"), r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
print("kernel,counter,mean_per_dispatch,dispatches")
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k},{c},{s / n:.6g},{n}")
