#!/usr/bin/env python3
"""Per-kernel means of every counter collected by tools/pmc_multi.sh: for each kernel the dispatches of its largest grid
(the finest level; the chunk-length trials of the first use are dropped by taking the most frequent grid among the long
ones).  Usage: pmc_multi_table.py <dir>"""
import csv, glob, sys
from collections import defaultdict
disp = defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    pas = f.split("/p")[-2] if "/p" in f else f
    for row in csv.DictReader(open(f)):
        g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
        disp[(row["Kernel_Name"], g, f, row.get("Dispatch_Id"))][row["Counter_Name"]] = float(row["Counter_Value"])
kern = defaultdict(lambda: defaultdict(list))  # kernel -> grid -> [counter dicts]
for (k, g, f, d), c in disp.items():
    kern[k][g].append(c)
for k in sorted(kern, key=lambda k: -max(kern[k])):
    gmax = max(kern[k])
    big = [g for g in kern[k] if g >= 0.5 * gmax]
    g = max(big, key=lambda g: len(kern[k][g]))
    acc = defaultdict(list)
    for c in kern[k][g]:
        for n, v in c.items():
            acc[n].append(v)
    m = {n: sum(v) / len(v) for n, v in acc.items()}
    print(f"== {k[:90]}  grid {g}  dispatches/pass ~{len(kern[k][g]) // 8}")
    for n in sorted(m):
        print(f"   {n:44s} {m[n]:18.1f}")
    w, wc = m.get("SQ_WAVES"), m.get("SQ_WAVE_CYCLES")
    if w and wc:
        per = lambda n: m.get(n, 0) / w
        print(f"   -- per wave: cycles {4 * wc / w:.0f}  VALU {per('SQ_INSTS_VALU'):.0f}  SALU {per('SQ_INSTS_SALU'):.0f}  VMEM rd {per('SQ_INSTS_VMEM_RD'):.0f} wr {per('SQ_INSTS_VMEM_WR'):.0f}  LDS {per('SQ_INSTS_LDS'):.0f}  branch {per('SQ_INSTS_BRANCH'):.0f}")
        sh = lambda n: 100 * m.get(n, 0) / wc
        print(f"   -- share of wave cycles: wait_any {sh('SQ_WAIT_ANY'):.1f}%  wait_inst_any {sh('SQ_WAIT_INST_ANY'):.1f}%  wait_inst_lds {sh('SQ_WAIT_INST_LDS'):.1f}%  active_any {sh('SQ_ACTIVE_INST_ANY'):.1f}%  valu {sh('SQ_ACTIVE_INST_VALU'):.1f}%  sca {sh('SQ_ACTIVE_INST_SCA'):.1f}%  lds {sh('SQ_ACTIVE_INST_LDS'):.1f}%  vmem {sh('SQ_ACTIVE_INST_VMEM'):.1f}%  misc {sh('SQ_ACTIVE_INST_MISC'):.1f}%")
    if m.get("TCC_REQ_sum"):
        print(f"   -- L2: hit rate {100 * m.get('TCC_HIT_sum', 0) / max(1, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0)):.1f}%  EA rd req {m.get('TCC_EA0_RDREQ_sum', 0):.0f} (32B: {m.get('TCC_EA0_RDREQ_32B_sum', 0):.0f})  mean outstanding EA reads {m.get('TCC_EA0_RDREQ_LEVEL_sum', 0) / max(1, m.get('TCC_CYCLE_sum', 1)) * 16:.1f} per channel-cycle x16")
    if m.get("TCP_TCC_READ_REQ_sum"):
        print(f"   -- TCP: mean L2 read latency {m.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / m['TCP_TCC_READ_REQ_sum']:.0f} cycles; pending-stall cycles {m.get('TCP_PENDING_STALL_CYCLES_sum', 0):.0f}")
