#!/bin/bash
# tools/mapping_sequence.sh <tag> [frames]: kernel sequence (start offset, duration) of ONE hipGraph-replayed mapping iteration
# of the synthetic TUM-like run (the last one: the window holds frames/5 + 1 keyframes) -> gpurun_out/<tag>_mapping_replay_kernel_sequence.txt
tag=${1:-map}; frames=${2:-16}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/pm_$tag -o s -- python3 $R/tools/slam_bench.py --config ${CFG:-tum} --graph --frames $frames > $O/${tag}_slam_trace_run.log 2>&1
python3 $R/tools/kernel_sequence.py $(find /tmp/pm_$tag -name '*kernel_trace.csv' | head -1) adam_kernel -1 > $O/${tag}_mapping_replay_kernel_sequence.txt 2>&1
tail -3 $O/${tag}_mapping_replay_kernel_sequence.txt
