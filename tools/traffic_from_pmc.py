#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md 'HBM' prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of 16-byte-per-lane reads, so it is doubled; WRITE_SIZE is exact for float atomics and
wide stores.  usage: traffic_from_pmc.py <fetch.csv> <write.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"mgs::(\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "correction": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
for k in sorted(set(fetch) | set(write)):
    out[k] = {"fetch_kib_raw": round(fetch.get(k, 0), 1), "write_kib_raw": round(write.get(k, 0), 1),
              "hbm_bytes": int(2 * fetch.get(k, 0) * 1024 + write.get(k, 0) * 1024)}
out["blend_backward_bytes_per_launch"] = (out.get("blend_backward_s_kernel") or out.get("blend_backward_t_kernel") or out.get("blend_backward_kernel") or {}).get("hbm_bytes")
# identity of the kernel sources these counters were collected on (bench.py ignores the file when it does not match)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402
out["csrc_sha256"] = csrc_hash()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
