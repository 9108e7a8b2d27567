#!/bin/bash
# tools/ab_blend.sh default v1 v2 ...: blend forward / backward stage times of kernel variants (monogs_amd/lib/variants/libmgs_<name>.so;
# "default" = the in-tree library) at C5 and at 100 k Gaussians / VGA, 1000 steps each.  A step that fails ends the call.
for v in "$@"; do
  if [ "$v" = default ]; then unset MGS_LIB_PATH; else export MGS_LIB_PATH=$PWD/monogs_amd/lib/variants/libmgs_$v.so; fi
  timeout -k 10 100 python bench.py --no-slam --no-cpu-baseline --steps 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v c5 ', d['value'], s['blend_fwd_ms'], s['blend_bwd_ms'])" || exit 1
  timeout -k 10 100 python bench.py --no-slam --no-cpu-baseline --steps 1000 --gaussians 100000 --intrinsics fr3_office 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v vga', d['value'], s['blend_fwd_ms'], s['blend_bwd_ms'])" || exit 1
done
