#!/usr/bin/env bash
# Build lpi_amd/csrc/variants/liblpi_hip_<suffix>.so with extra compiler flags on SOME sources (ablation / A-B builds; load it with LPI_LIB=<path> or
# tools/gemm_variant.py <suffix>):   tools/build_variant.sh <suffix> <source[,source...] without .hip> <flags...>
# A variant reports lpi_version() = ABI + 1 000 000 (api.hip is always recompiled with -DLPI_VARIANT_BUILD): the binding loads it only through LPI_LIB and
# says so.  variants/ is git-ignored but travels to the GPU box with every gpurun push: delete it after the A/B (tools/build_variant.sh --clean).
set -euo pipefail
cd "$(dirname "$0")/../lpi_amd/csrc"
if [ "${1:-}" = "--clean" ]; then rm -rf variants; echo "removed $(pwd)/variants"; exit 0; fi
suffix=$1; srcs=",$2,"; shift 2
mkdir -p variants/obj
objs=""
for f in api gemm gemm256 gemm256p gemm256x128 gemm_rows attention attention4 attn_pooled attn_stream attn_long rowops loss interact bpe host; do
  if [[ "$srcs" == *",$f,"* ]] || [ $f = api ]; then
    extra=""
    case $f in attention|attention4) extra="-fno-honor-nans";; api) extra="-DLPI_VARIANT_BUILD";; esac
    if [[ "$srcs" == *",$f,"* ]]; then flags=("$@"); else flags=(); fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $extra "${flags[@]}" -c $f.hip -o variants/obj/${f}_$suffix.o &
    objs="$objs variants/obj/${f}_$suffix.o"
  else
    objs="$objs build/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/liblpi_hip_$suffix.so $objs -lpthread
echo "built $(pwd)/variants/liblpi_hip_$suffix.so"
