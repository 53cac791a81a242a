#!/usr/bin/env python3
"""One steady-state V-cycle from a rocprofv3 kernel trace: every launch with its duration and the idle gap in front of
it -- where the cycle's time below the finest level goes.  python tools/cycle_timeline.py kernel_trace.csv [cycle-from-end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
# a cycle ends with the fold of the norm's partial sums
ends = [i for i, r in enumerate(rows) if name(r).startswith("fold_partials") or name(r).startswith("fold2_partials")]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
t_prev = int(rows[lo - 1]["End_Timestamp"])
t0 = t_prev
tot_k = tot_g = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r["Grid_Size"]) if "Grid_Size" in r else 0
    wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 1
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - t_prev) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  blocks {g // max(wg, 1):6d}  {name(r)}")
    tot_k += e - s
    tot_g += s - t_prev
    t_prev = e
print(f"cycle: {(t_prev - t0) / 1e3:.1f} us = kernels {tot_k / 1e3:.1f} + gaps {tot_g / 1e3:.1f}; launches {hi - lo}")
