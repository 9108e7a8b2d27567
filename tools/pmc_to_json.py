#!/usr/bin/env python3
"""SQ counters per kernel (mean over dispatches) from one or more rocprofv3 --pmc counter CSVs -> a JSON file stamped
with the hash of the kernel sources (bench.py attaches `roofline.valu` only when the stamp matches the sources it runs).
usage: pmc_to_json.py <out.json> <counter_collection.csv> [more.csv ...]"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"mgs::(\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"unit": "counter value per launch (mean over the dispatches of the profiled bench.py run, C5 workload)",
       "csrc_sha256": csrc_hash()}
for k, cs in sorted(acc.items()):
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: out[k] for k in out if "blend" in k}, indent=1))
