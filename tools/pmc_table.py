#!/usr/bin/env python3
"""Pivot a rocprofv3 *_counter_collection.csv into one row per kernel (mean over dispatches).
usage: pmc_table.py <counter_collection.csv> [kernel-substring ...]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    m = re.search(r"(radix_sort_onesweep_\w+|mgs::\w+)", name)
    if not m:
        continue
    k = m.group(1)
    if filt and not any(f in k for f in filt):
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")
