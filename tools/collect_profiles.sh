#!/bin/bash
# Everything under profiles/ that bench.py's roofline record points at, collected in one go on the GPU box
# (run from the repo root: tools/collect_profiles.sh r02).  Output in gpurun_out/<tag>_*; copy what is to be judged
# into profiles/.  rocprofv3 runs python3 itself (no wrapper hop), counters in passes of their own (no --stats with --pmc).
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-slam"
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
# 1. per-kernel time (the same command bench.py's live HIP-event figures come from)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/ks -o b -- $B --steps 20 > $O/${tag}_rocprof_run.log 2>&1
python3 $R/tools/summarize_prof.py $(find /tmp/prof_$tag/ks -name '*kernel_stats.csv' | head -1) $O/${tag}_bench_kernel_stats.csv >> $O/${tag}_rocprof_run.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE need a pass each (TCC slots)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_$tag/pf -o f -- $B --steps 3 --warmup 1 > /tmp/prof_$tag/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_$tag/pw -o w -- $B --steps 3 --warmup 1 > /tmp/prof_$tag/pw.log 2>&1
python3 $R/tools/traffic_from_pmc.py $(find /tmp/prof_$tag/pf -name '*counter_collection.csv' | head -1) $(find /tmp/prof_$tag/pw -name '*counter_collection.csv' | head -1) $O/${tag}_traffic.json > $O/${tag}_pmc_traffic.txt 2>&1
# 3. SQ counters (8 per pass)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_$tag/sa -o a -- $B --steps 3 --warmup 1 > /tmp/prof_$tag/sa.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d /tmp/prof_$tag/sb -o b -- $B --steps 3 --warmup 1 > /tmp/prof_$tag/sb.log 2>&1
python3 $R/tools/pmc_to_json.py $O/${tag}_pmc_valu.json $(find /tmp/prof_$tag/sa /tmp/prof_$tag/sb -name '*counter_collection.csv') > $O/${tag}_pmc_sq.txt 2>&1
# 3b. issue cost per instruction class (micro-benchmark) and the static mix of the backward's hot loop
$R/tools/ubench/valu_rate > $O/${tag}_ubench_valu_rates.txt 2>&1
grep '^JSON ' $O/${tag}_ubench_valu_rates.txt | cut -c6- > $O/${tag}_valu_costs.json
python3 $R/tools/isa_mix.py $O/${tag}_isa_mix.json > /dev/null 2>&1
# 3c. the radix sort alone (tools/ubench/sort_bench: against a stable CPU sort, whole-sort time, phase trace of a tile)
( for m in depth depthfar tile few; do $R/tools/ubench/sort_bench $m 0; done
  for n in 40000 100000 415000 1000000; do $R/tools/ubench/sort_bench depth 0 $n; $R/tools/ubench/sort_bench tile 0 $n; done ) > $O/${tag}_sort_bench.txt 2>&1
# 3d. round 5: the single-workgroup depth chain alone (phase stamps), the group-stream counts, the tail probe
( for n in 10000 20000 24576; do $R/tools/ubench/depth_small_bench $n; done ) > $O/${tag}_small_depth_chain_bench.txt 2>&1
python3 $R/tools/group_stats.py > $O/${tag}_group_stream_counts.txt 2>&1
python3 $R/tools/tail_probe.py 30 > $O/${tag}_tail_probe.txt 2>&1
# 4. the plain line (with the stamped files in place the roofline record carries traffic and valu)
cp $O/${tag}_traffic.json $R/profiles/traffic.json; cp $O/${tag}_pmc_valu.json $R/profiles/pmc_valu.json
cp $O/${tag}_valu_costs.json $R/profiles/valu_costs.json; cp $O/${tag}_isa_mix.json $R/profiles/isa_mix.json
python3 $R/bench.py > $O/${tag}_bench_default.log 2>&1
tail -1 $O/${tag}_bench_default.log | cut -c1-2500
