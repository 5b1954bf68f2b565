#!/usr/bin/env bash
# Build liblpi_hip_<suffix>.so with extra -D flags on ONE source (ablation / A-B builds for tools/gemm_variant.py):
#   tools/build_variant.sh <suffix> <source-without-.hip> <flags...>
set -euo pipefail
cd "$(dirname "$0")/../lpi_amd/csrc"
suffix=$1; src=$2; shift 2
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c $src.hip -o build_var/${src}_$suffix.o
objs=""
for f in api gemm gemm256 gemm256p gemm256x128 gemm_duo attention attention2 attn_pooled rowops loss bpe; do
  if [ "$f" = "$src" ]; then objs="$objs build_var/${src}_$suffix.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblpi_hip_$suffix.so $objs
echo "built liblpi_hip_$suffix.so"
