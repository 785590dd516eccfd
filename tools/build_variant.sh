#!/bin/bash
# Builds vision_amd/libexp_<name>.so: the library with ONE source replaced by a variant file (for tools/ab_*.py A/B runs).
#   tools/build_variant.sh <name> <source-in-csrc, e.g. nms.hip> <variant file>
set -e
name=$1; src=$2; var=$(realpath "$3")
cd "$(dirname "$0")/../vision_amd/csrc"
FLAGS="$EXTRA_FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
cp "$var" /tmp/variant_$name.hip
/opt/rocm/bin/hipcc $FLAGS -I. -c -x hip /tmp/variant_$name.hip -o /tmp/variant_$name.o
objs=""
for f in abi iou nms decode roipool roipool_generic kmeans loss evalap comm; do
  if [ "$f.hip" == "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libexp_$name.so $objs -ldl
echo built vision_amd/libexp_$name.so
