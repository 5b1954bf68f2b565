#!/usr/bin/env bash
# compile one csrc source for gfx950 into /tmp and list the kernels whose private segment (scratch: register spills) is not empty
#   usage: tools/spill_check.sh gemm256p [extra hipcc flags]
set -euo pipefail
f=$1; shift
cd "$(dirname "$0")/../lpi_amd/csrc"
d=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c $f.hip -o $d/$f.o
cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $f.o > /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $d/*amdgcn* | awk '/\.name:/{n=$2} /\.private_segment_fixed_size:/{if ($2 != 0) print n, $2} /\.vgpr_count:/{v[n]=$2}' | sort | uniq
echo "(listed: kernels with scratch; none listed = no spills)"
rm -rf $d
