#!/bin/bash
# The radix sort on the GPU box: tools/ubench/sort_bench on the C5 shapes and a few SLAM-sized ones (against a stable CPU
# sort, with the phase trace of a tile), then the sort / parity / feature tests and the C5 and VGA stage times.
for m in depth depthfar tile few; do timeout -k 5 60 tools/ubench/sort_bench $m 0 || exit 1; done
for n in 40000 100000 415000 1000000; do timeout -k 5 60 tools/ubench/sort_bench depth 0 $n | head -2; timeout -k 5 60 tools/ubench/sort_bench depthfar 0 $n | head -2; timeout -k 5 60 tools/ubench/sort_bench tile 0 $n | head -2; done
python -m pytest tests/test_gpu_sort.py tests/test_gpu_parity.py tests/test_gpu_features.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-slam --no-cpu-baseline --steps 300 2>> gpurun_out/abr.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('c5', d['value'], s)"
python bench.py --no-slam --no-cpu-baseline --steps 300 --gaussians 100000 --intrinsics fr3_office 2>> gpurun_out/abr.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('vga', d['value'], s)"
