#!/usr/bin/env python3
"""Mean duration of the finest-level launches of each kernel in a rocprofv3 kernel trace (csv): per kernel name the
dispatch group with the largest grid.  Usage: finest_from_trace.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0) * max(1, int(row.get("Grid_Size_Y") or 1)) * max(1, int(row.get("Grid_Size_Z") or 1))
    acc[(row["Kernel_Name"], g)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
best = {}
for (k, g), v in acc.items():
    tot = sum(v)
    if k not in best or sum(best[k][1]) / len(best[k][1]) < tot / len(v):
        best[k] = (g, v)
print("Finest-level (largest mean duration) dispatch group of each kernel in the kernel trace of `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline`:")
for k, (g, v) in sorted(best.items(), key=lambda t: -sum(t[1][1])):
    print(f"  {k[:72]:72s} grid {g:9d}  launches {len(v):4d}  mean {sum(v) / len(v) / 1e6:8.4f} ms  min {min(v) / 1e6:8.4f}  max {max(v) / 1e6:8.4f}")
