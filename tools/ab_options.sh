#!/bin/bash
# A/B of mgs_debug_set_option knobs on the GPU box: tools/ab_options.sh "" "blend_wgs_per_cu=3" ...  Prints the stage
# times at C5 and at 100 k Gaussians / VGA and the captured tracking / mapping rates of the synthetic TUM-like run.
for o in "$@"; do
  export MGS_DEBUG_OPTIONS="$o"
  python bench.py --no-slam --no-cpu-baseline --steps 200 2>> gpurun_out/abo.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$o] c5 ', d['value'], d['stages_ms'])"
  python bench.py --no-slam --no-cpu-baseline --steps 200 --gaussians 100000 --intrinsics fr3_office 2>> gpurun_out/abo.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$o] vga', d['value'], d['stages_ms'])"
  python tools/slam_bench.py --config tum --graph 2>> gpurun_out/abo.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$o] slam', round(d['tracking_steady_iters_per_s']), round(d['mapping_steady_iters_per_s']))"
done
