#!/bin/bash
# tools/pmc_clock.sh <tag>: GRBM_GUI_ACTIVE (core clocks while the GPU is busy) per kernel, for the FMA micro-benchmark and for
# one bench.py run -> the core clock each kernel actually ran at (cycles / duration).  gpurun_out/clock_<tag>.txt
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_a /tmp/clk_b
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_a -o c -- $R/tools/ubench/valu_rate > /tmp/clk_a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_b -o c -- python3 $R/bench.py --no-cpu-baseline --no-slam --steps 3 --warmup 1 > /tmp/clk_b.log 2>&1
python3 - "$R/gpurun_out/clock_$tag.txt" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
for d in ("/tmp/clk_a", "/tmp/clk_b"):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(d, "no output", file=out); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        ns, name = dur.get(r["Dispatch_Id"], (0, r["Kernel_Name"]))
        a = agg[name[:70]]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
    for k, (cy, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        if ns > 0:
            print(f"{k:70s} n={n:4d} avg_us={ns / n / 1e3:9.1f} cycles/ns(GHz, summed over XCDs?)={cy / ns:8.3f}", file=out)
out.close()
PY
cat $R/gpurun_out/clock_$tag.txt
