#!/usr/bin/env python3
"""Mean duration of the finest-level launches of each kernel in a rocprofv3 kernel trace (csv): per kernel name the
steady-state dispatch group of the finest level.  Usage: finest_from_trace.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0) * max(1, int(row.get("Grid_Size_Y") or 1)) * max(1, int(row.get("Grid_Size_Z") or 1))
    acc[(row["Kernel_Name"], g)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
# per kernel: the launches of the finest level (at least half as long as the kernel's longest launch -- grids of different
# levels can coincide), and among their grids the one launched most often: the steady-state configuration.  The few
# launches with other grids are the chunk lengths the sweep launcher tries the first time a shape meets a level.
best = {}
for k in {k for (k, g) in acc}:
    top = max(max(v) for (kk, g), v in acc.items() if kk == k)
    groups = {g: [x for x in v if x >= 0.5 * top] for (kk, g), v in acc.items() if kk == k}
    g = max((g for g, v in groups.items() if v), key=lambda g: (len(groups[g]), -g))
    best[k] = (g, groups[g])
print("Finest-level steady-state dispatch group (most launches among the long ones) of each kernel in the kernel trace of `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline`:")
for k, (g, v) in sorted(best.items(), key=lambda t: -sum(t[1][1])):
    print(f"  {k[:72]:72s} grid {g:9d}  launches {len(v):4d}  mean {sum(v) / len(v) / 1e6:8.4f} ms  min {min(v) / 1e6:8.4f}  max {max(v) / 1e6:8.4f}")
