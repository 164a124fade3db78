#!/bin/bash
# Build a kernel variant of libmpe_hip.so without touching the tree:
#   tools/build_variant.sh NAME [-DFOO=1 ...] [--sed 's/a/b/' FILE ...]
# copies csrc/ + include/ to build_variants/NAME/, applies the sed edits (FILE relative to csrc/ or include/), builds
# with the extra flags, leaves build_variants/NAME/libmpe_hip.so (git-ignored, travels with gpurun).
# Use: MPE_LIB=build_variants/NAME/libmpe_hip.so python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$R/build_variants/$NAME
rm -rf "$D"; mkdir -p "$D/pkg/csrc" "$D/include"
cp $R/rpg_monocular_pose_estimator_amd/csrc/*.{hip,h,cpp} $R/rpg_monocular_pose_estimator_amd/csrc/Makefile "$D/pkg/csrc/"
cp $R/include/*.h "$D/include/"
FLAGS=""
while [ $# -gt 0 ]; do
  case "$1" in
    --sed) expr=$2; f=$3; shift 3
           if [ -f "$D/pkg/csrc/$f" ]; then sed -i "$expr" "$D/pkg/csrc/$f"; else sed -i "$expr" "$D/include/$f"; fi ;;
    *) FLAGS="$FLAGS $1"; shift ;;
  esac
done
# (the Makefile refers to ../../include and writes ../libmpe_hip.so: pkg/csrc sits two levels below $D for that)
make -s -C "$D/pkg/csrc" -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $FLAGS" 2>&1 | grep -v "^$" || true
mv "$D/pkg/libmpe_hip.so" "$D/libmpe_hip.so"
rm -f "$D"/pkg/csrc/*.o
echo "built $D/libmpe_hip.so"
