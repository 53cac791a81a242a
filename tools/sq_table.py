#!/usr/bin/env python3
"""Per-kernel SQ counter table from one rocprofv3 --pmc pass (csv): for each kernel name the dispatch group with the
most wave-cycles (= the finest level), means per launch.  Usage: sq_table.py <dir>"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
        acc[(row["Kernel_Name"], g)][row["Counter_Name"]].append(float(row["Counter_Value"]))
best = {}
for (k, g), cs in acc.items():
    wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(cs.get("SQ_WAVE_CYCLES", [0])))
    if k not in best or wc > best[k][0]:
        best[k] = (wc, g, cs)
for k, (wc, g, cs) in sorted(best.items(), key=lambda t: -t[1][0]):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    waves = m.get("SQ_WAVES", 1)
    print(f"{k[:58]:58s} grid {g:8d} waves {waves:7.0f} | per wave: cycles {4 * wc / waves:9.0f} VALU {m.get('SQ_INSTS_VALU', 0) / waves:7.0f} "
          f"SALU {m.get('SQ_INSTS_SALU', 0) / waves:7.0f} | wait {100 * m.get('SQ_WAIT_ANY', 0) / max(wc, 1):4.1f}% "
          f"stall {100 * m.get('SQ_WAIT_INST_ANY', 0) / max(wc, 1):4.1f}% active {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / max(wc, 1):4.1f}% "
          f"(VALU {100 * m.get('SQ_ACTIVE_INST_VALU', 0) / max(wc, 1):4.1f}% SCA {100 * m.get('SQ_ACTIVE_INST_SCA', 0) / max(wc, 1):4.1f}%)")
