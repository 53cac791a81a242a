#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (tools/pmc.sh): dir A with FETCH_SIZE, dir B with
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum.  Per kernel name the dispatch group with the largest traffic (= the finest
level) is averaged.  gfx950 correction: read bytes = 2 x FETCH_SIZE for 16-byte-per-lane loads (see the header of
profiles/r01_pmc_traffic_finest_level.txt).  Usage: pmc_traffic_table.py <dirA> <dirB>"""
import csv, glob, os, sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))  # kernel -> grid -> counter -> values
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):  # (rocprofv3 nests a host-name directory)
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][(int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row["Grid_Size"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


A, B = load(sys.argv[1]), load(sys.argv[2])
for k in sorted(A, key=lambda k: -max(sum(v) / len(v) for (g, c), v in A[k].items() if c == "FETCH_SIZE")):
    best = max(((g, sum(v) / len(v), len(v)) for (g, c), v in A[k].items() if c == "FETCH_SIZE"), key=lambda t: t[1])
    g, fetch_kb, n = best
    w = B.get(k, {})
    wr = w.get((g, "WRITE_SIZE"), [0.0])
    hit, miss = w.get((g, "TCC_HIT_sum"), [0.0]), w.get((g, "TCC_MISS_sum"), [0.0])
    wr_kb = sum(wr) / len(wr)
    rd_gb, wr_gb = 2 * fetch_kb * 1024 / 1e9, wr_kb * 1024 / 1e9
    print(f"{k[:70]:70s} | grid {g:9d} | n={n:3d} | FETCH_SIZE {fetch_kb:11.1f} KB (x2 -> {rd_gb:6.3f} GB read) | "
          f"WRITE_SIZE {wr_kb:11.1f} KB ({wr_gb:6.3f} GB) | L2 hit {sum(hit) / len(hit):.4g} miss {sum(miss) / len(miss):.4g} | "
          f"HBM-side bytes/launch {rd_gb + wr_gb:6.3f} GB")
