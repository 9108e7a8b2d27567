#!/bin/bash
# tools/ab_bwd.sh "opts1" "opts2" ...: blend forward / backward stage times at C5 and at 100 k / VGA for mgs_debug_set_option
# settings (MGS_DEBUG_OPTIONS syntax, "" = defaults), 600 steps each; optional MGS_LIB_PATH for variant libraries.
for o in "$@"; do
  export MGS_DEBUG_OPTIONS="$o"
  timeout -k 10 120 python bench.py --no-slam --no-cpu-baseline --steps 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('[$o] c5 ', d['value'], s['blend_fwd_ms'], s['blend_bwd_ms'], d['roofline']['avg_ms'])" || exit 1
  timeout -k 10 120 python bench.py --no-slam --no-cpu-baseline --steps 600 --gaussians 100000 --intrinsics fr3_office 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('[$o] vga', d['value'], s['blend_fwd_ms'], s['blend_bwd_ms'], d['roofline']['avg_ms'])" || exit 1
done
