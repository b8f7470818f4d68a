#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per (kernel, grid size, counter): mean per dispatch.
Keyed by GRID SIZE as well as by name: the three level launches of pyr_level_kernel<true> share a name, and legs at other
batch sizes must not pollute the figure of the launch bench.py's roofline refers to."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            m = re.search(r"(klt_kernel|pyr_level_kernel<\w+>|pyr_down_l0_kernel|pyr_tail_kernel|pyr_border_kernel|ekf_\w+?_kernel|vu_\w+?_kernel(?:_2percu)?|gftt_\w+?_kernel|rot_ransac_kernel)", name)
            grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            k = (m.group(1) if m else name[:48].replace(",", ";"), grid, r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
print("kernel,grid,counter,mean_per_dispatch,dispatches")
for (k, g, c), (s, n) in sorted(acc.items()):
    print(f"{k},{g},{c},{s / n:.6g},{n}")
