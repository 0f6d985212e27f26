#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compile the unmodified HIP sources for the host against
# the SIMT emulator header (tests/hipemu/hip/hip_runtime.h).  See that header.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/hipemu/_build
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=g++
$CXX -x c++ -std=c++17 -O2 -g0 -fPIC -shared -pthread -Wno-unused-value -I tests/hipemu \
    deft_amd/csrc/igemm.hip deft_amd/csrc/igemm3.hip deft_amd/csrc/dcn.hip deft_amd/csrc/direct.hip deft_amd/csrc/ops.hip deft_amd/csrc/assoc.hip -o tests/hipemu/_build/libdeft_emu.so
echo built tests/hipemu/_build/libdeft_emu.so
