#!/usr/bin/env python3
"""Per-kernel SQ counter table from one rocprofv3 --pmc pass (csv): for each kernel name the finest level's
steady-state dispatch group (most launches among the long ones), means per launch.  Usage: sq_table.py <dir>"""
import csv, glob, sys
from collections import defaultdict
disp = defaultdict(dict)  # (kernel, grid, dispatch id) -> {counter: value}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
        disp[(row["Kernel_Name"], g, row.get("Dispatch_Id"))][row["Counter_Name"]] = float(row["Counter_Value"])
# the finest level's launches (at least half the kernel's largest wave-cycle count -- grids of different levels can
# coincide) and among their grids the one launched most often: the steady state (the others are the chunk lengths the
# sweep launcher tries the first time a shape meets a level)
best = {}
for k in {k for (k, g, d) in disp}:
    mine = {(g, d): c for (kk, g, d), c in disp.items() if kk == k}
    top = max(c.get("SQ_WAVE_CYCLES", 0.) for c in mine.values())
    groups = defaultdict(list)
    for (g, d), c in mine.items():
        if c.get("SQ_WAVE_CYCLES", 0.) >= 0.5 * top:
            groups[g].append(c)
    g = max(groups, key=lambda g: (len(groups[g]), g))
    cs = defaultdict(list)
    for c in groups[g]:
        for name, v in c.items():
            cs[name].append(v)
    best[k] = (sum(cs["SQ_WAVE_CYCLES"]) / max(1, len(cs["SQ_WAVE_CYCLES"])), g, cs)
for k, (wc, g, cs) in sorted(best.items(), key=lambda t: -t[1][0]):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    waves = m.get("SQ_WAVES", 1)
    print(f"{k[:58]:58s} grid {g:8d} waves {waves:7.0f} | per wave: cycles {4 * wc / waves:9.0f} VALU {m.get('SQ_INSTS_VALU', 0) / waves:7.0f} "
          f"SALU {m.get('SQ_INSTS_SALU', 0) / waves:7.0f} | wait {100 * m.get('SQ_WAIT_ANY', 0) / max(wc, 1):4.1f}% "
          f"stall {100 * m.get('SQ_WAIT_INST_ANY', 0) / max(wc, 1):4.1f}% active {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / max(wc, 1):4.1f}% "
          f"(VALU {100 * m.get('SQ_ACTIVE_INST_VALU', 0) / max(wc, 1):4.1f}% SCA {100 * m.get('SQ_ACTIVE_INST_SCA', 0) / max(wc, 1):4.1f}%)")
