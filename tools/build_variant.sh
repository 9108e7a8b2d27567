#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-DFLAG ...]: rebuilds ONE translation unit with extra flags and links a
# variant library monogs_amd/lib/variants/libmgs_<name>.so next to the in-tree one (kernel A/B experiments).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
n=$1; src=$2; shift 2
base=${src%.hip}
mkdir -p $ROOT/monogs_amd/lib/variants
extra=""
[ "$base" = blend ] && extra="-fno-slp-vectorize -Wno-inline-asm"
[ "$base" = preprocess ] && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -DNDEBUG $extra "$@" \
  -c $ROOT/monogs_amd/csrc/$src -o $ROOT/monogs_amd/lib/variants/${base}_$n.o
objs=""
for o in api preprocess binning radix_sort blend knn losses pose optim; do
  if [ "$o" = "$base" ]; then objs="$objs $ROOT/monogs_amd/lib/variants/${base}_$n.o"; else objs="$objs $ROOT/monogs_amd/lib/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/monogs_amd/lib/variants/libmgs_$n.so $objs
echo built libmgs_$n.so
