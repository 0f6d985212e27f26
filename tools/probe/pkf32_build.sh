#!/bin/bash
# builds tools/probe/pkf32_min.bin (plain HIP, no torch): one translation unit with packed fp32 allowed, one without
set -e
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
$H --offload-arch=gfx950 -O3 -std=c++17 -DPK_TU=1 -c pkf32_min.hip -o /tmp/pkf32_pk.o
$H --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -DPK_TU=0 -c pkf32_min.hip -o /tmp/pkf32_nopk.o 2> >(grep -v "packed-fp32-ops" >&2)
$H --offload-arch=gfx950 /tmp/pkf32_pk.o /tmp/pkf32_nopk.o -o pkf32_min.bin
echo built tools/probe/pkf32_min.bin
