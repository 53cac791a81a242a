#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter over its dispatches."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(d, "|", k, "|", " ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))
