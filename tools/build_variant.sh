#!/usr/bin/env bash
# Build liblpi_hip_<suffix>.so with extra compiler flags on SOME sources (ablation / A-B builds; load it with LPI_LIB=<path> or
# tools/gemm_variant.py <suffix>):   tools/build_variant.sh <suffix> <source[,source...] without .hip> <flags...>
set -euo pipefail
cd "$(dirname "$0")/../lpi_amd/csrc"
suffix=$1; srcs=",$2,"; shift 2
mkdir -p build_var
objs=""
for f in api gemm gemm256 gemm256p gemm256x128 attention attention4 attn_pooled rowops loss interact bpe; do
  if [[ "$srcs" == *",$f,"* ]]; then
    extra=""
    case $f in attention|attention4) extra="-fno-honor-nans";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $extra "$@" -c $f.hip -o build_var/${f}_$suffix.o &
    objs="$objs build_var/${f}_$suffix.o"
  else
    objs="$objs build/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblpi_hip_$suffix.so $objs
echo "built $(pwd)/liblpi_hip_$suffix.so"
