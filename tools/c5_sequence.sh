#!/bin/bash
# tools/c5_sequence.sh <tag>: kernel sequence (start offset, duration) of ONE C5 step (2 M Gaussians, 1080p, fwd+bwd) out of a
# rocprofv3 kernel trace of bench.py -> gpurun_out/<tag>_c5_step_kernel_sequence.txt
tag=${1:-seq}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5s_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5s_$tag -o s -- python3 $R/bench.py --no-cpu-baseline --no-slam --steps 8 --warmup 3 --profile-steps 0 > $O/${tag}_c5_trace_run.log 2>&1
python3 $R/tools/kernel_sequence.py $(find /tmp/c5s_$tag -name '*kernel_trace.csv' | head -1) tau_finalize_kernel ${2:-6} > $O/${tag}_c5_step_kernel_sequence.txt 2>&1
cat $O/${tag}_c5_step_kernel_sequence.txt
