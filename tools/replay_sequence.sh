#!/bin/bash
# tools/replay_sequence.sh <tag>: kernel sequence of ONE hipGraph-replayed tracking iteration (40 k Gaussians / VGA) out of a
# rocprofv3 kernel trace of the synthetic TUM-like run -> gpurun_out/<tag>_tracking_replay_kernel_sequence.txt
tag=${1:-seq}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_$tag -o s -- python3 $R/tools/slam_bench.py --config ${CFG:-tum} --graph --frames 6 > $O/${tag}_slam_trace_run.log 2>&1
python3 $R/tools/kernel_sequence.py $(find /tmp/ps_$tag -name '*kernel_trace.csv' | head -1) pose_step_kernel -1 "blend_backward_s_kernel<true>" > $O/${tag}_tracking_replay_kernel_sequence.txt 2>&1
tail -30 $O/${tag}_tracking_replay_kernel_sequence.txt
