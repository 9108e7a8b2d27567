#!/usr/bin/env python3
"""Shorten a rocprofv3 *_kernel_stats.csv (template-heavy names) into a readable summary.
usage: summarize_prof.py <kernel_stats.csv> <out.csv>"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(radix_sort_onesweep_\w+|mgs::\w+|at::native::\w+|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:80]


rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, dict(Calls=0, TotalDurationNs=0, MinNs=1 << 62, MaxNs=0))
    a["Calls"] += int(r["Calls"])
    a["TotalDurationNs"] += int(r["TotalDurationNs"])
    a["MinNs"] = min(a["MinNs"], int(r["MinNs"]))
    a["MaxNs"] = max(a["MaxNs"], int(r["MaxNs"]))
tot = sum(a["TotalDurationNs"] for a in agg.values())
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["TotalDurationNs"]):
        w.writerow([k, a["Calls"], a["TotalDurationNs"], round(a["TotalDurationNs"] / a["Calls"], 1),
                    round(100 * a["TotalDurationNs"] / tot, 2), a["MinNs"], a["MaxNs"]])
