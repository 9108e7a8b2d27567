#!/bin/bash
# A/B of kernel variants on the GPU box: tools/ab_variants.sh default fm1 fm2 ...  (variant libraries under
# monogs_amd/lib/variants/libmgs_<name>.so; "default" = the in-tree library).  Prints the stage times at C5 and at
# 100 k Gaussians / VGA.
for v in "$@"; do
  if [ "$v" = default ]; then unset MGS_LIB_PATH; else export MGS_LIB_PATH=$PWD/monogs_amd/lib/variants/libmgs_$v.so; fi
  python bench.py --no-slam --no-cpu-baseline 2>> gpurun_out/ab_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5 ', d['value'], d['stages_ms'])"
  python bench.py --no-slam --no-cpu-baseline --gaussians 100000 --intrinsics fr3_office 2>> gpurun_out/ab_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v vga', d['value'], d['stages_ms'])"
done
