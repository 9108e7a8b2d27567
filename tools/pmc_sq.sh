#!/bin/bash
# SQ counter passes for the blend kernels on the C5 workload (run on the GPU box from the repo root):
#   tools/pmc_sq.sh <tag> [variant]      -> gpurun_out/pmc_<tag>_{a,b}.txt
# Two passes of 8 SQ counters (the per-pass limit on gfx950); rocprofv3 runs the program itself (no wrapper hop).
tag=$1; variant=$2
R=${GRAFT_REPO_ROOT:-$PWD}
if [ -n "$variant" ] && [ "$variant" != default ]; then export MGS_LIB_PATH=$R/monogs_amd/lib/variants/libmgs_$variant.so; fi
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES"
for p in a b; do
  if [ $p = a ]; then C="$A"; else C="$B"; fi
  rm -rf /tmp/pmc_${tag}_$p
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$p -o c -- python3 $R/bench.py --no-cpu-baseline --no-slam --steps 3 --warmup 1 > /tmp/pmc_${tag}_$p.log 2>&1
  f=$(find /tmp/pmc_${tag}_$p -name '*counter_collection.csv' | head -1)
  python3 $R/tools/pmc_table.py $f ${PMC_FILTER:-blend} > $R/gpurun_out/pmc_${tag}_$p.txt 2>&1 || tail -5 /tmp/pmc_${tag}_$p.log
done
cat $R/gpurun_out/pmc_${tag}_a.txt $R/gpurun_out/pmc_${tag}_b.txt
