#!/usr/bin/env python3
"""Print the kernel sequence between the last two launches of a marker kernel in a rocprofv3 kernel trace CSV."""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -1     # which occurrence ends the window (default: the last)
must = sys.argv[4] if len(sys.argv) > 4 else None      # optional: the window must contain a kernel with this substring
if must:                                                # ... then the k-th such window counts
    wins = [(idx[j - 1], idx[j]) for j in range(1, len(idx))
            if any(must in r["Kernel_Name"] for r in rows[idx[j - 1] + 1:idx[j] + 1])]
    a, b = wins[k]
else:
    a, b = idx[k - 1], idx[k]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
print(f"total {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
