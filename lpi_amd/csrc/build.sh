#!/usr/bin/env bash
# Build liblpi_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in api gemm gemm256 gemm256p gemm256x128 gemm_rows attention attention4 attn_pooled attn_stream attn_long rowops loss interact bpe host; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_epilogue.h -nt build/$f.o ] || [ attn_softmax.h -nt build/$f.o ] || [ unicode_ln.h -nt build/$f.o ] || [ gemm256_tile.h -nt build/$f.o ] || [ gemm256x128_tile.h -nt build/$f.o ] || [ ../../include/lpi_hip.h -nt build/$f.o ]; then
    extra=""
    case $f in attention|attention4) extra="-fno-honor-nans";; esac      # see attn_softmax.h (attn_max3)
    $HIPCC $FLAGS $extra -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o liblpi_hip.so build/api.o build/gemm.o build/gemm256.o build/gemm256p.o build/gemm256x128.o build/gemm_rows.o build/attention.o build/attention4.o build/attn_pooled.o build/attn_stream.o build/attn_long.o build/rowops.o build/loss.o build/interact.o build/bpe.o build/host.o -lpthread
echo "built $(pwd)/liblpi_hip.so"
