#!/bin/bash
# Stage times of the eager render + backward over a few scene shapes (looking for kernels that fall off a cliff):
# tools/sweep_stages.sh "N intrinsics" ...
for cfg in "$@"; do
  set -- $cfg
  python bench.py --no-slam --no-cpu-baseline --steps 100 --gaussians $1 --intrinsics $2 2>> gpurun_out/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'], d['stages_ms'])"
done
