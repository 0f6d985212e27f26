#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compile the unmodified HIP sources for the host against
# the SIMT emulator header (tests/hipemu/hip/hip_runtime.h).  See that header.
# Like libdeft_hip.so (deft_amd/build.py) the result carries BOTH arithmetics: the sources once with two fp16 pieces (entry points `name`)
# and once with three bf16 pieces, every symbol those objects define renamed to `name_p3` (llvm-objcopy --redefine-syms).
set -e
cd "$(dirname "$0")/../.."
B=tests/hipemu/_build
mkdir -p $B/o2 $B/o3
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=g++
OBJCOPY=/opt/rocm/lib/llvm/bin/llvm-objcopy
[ -x "$OBJCOPY" ] || OBJCOPY=objcopy
FLAGS="-x c++ -std=c++17 -O2 -g0 -fPIC -pthread -Wno-unused-value -I tests/hipemu"
pids=""
for f in igemm igemm3 dcn direct ops pairmlp assoc; do
    if [ ! -f $B/o2/$f.o ] || [ deft_amd/csrc/$f.hip -nt $B/o2/$f.o ] || [ deft_amd/csrc/common.h -nt $B/o2/$f.o ] || [ include/deft_hip.h -nt $B/o2/$f.o ] || [ tests/hipemu/hip/hip_runtime.h -nt $B/o2/$f.o ]; then
        X=""; [ $f = assoc ] && X="-ffp-contract=off"
        $CXX $FLAGS $X -c deft_amd/csrc/$f.hip -o $B/o2/$f.o & pids="$pids $!"
        [ $f = assoc ] || { $CXX $FLAGS -DDEFT_PIECES=3 -c deft_amd/csrc/$f.hip -o $B/o3/$f.o & pids="$pids $!"; }
    fi
done
for p in $pids; do wait $p; done
: > $B/redefine.map
for f in igemm igemm3 dcn direct ops pairmlp; do nm --defined-only --extern-only $B/o3/$f.o | awk 'NF>=3 {print $NF}'; done | sort -u | awk '{print $1, $1 "_p3"}' > $B/redefine.map
TW=""
for f in igemm igemm3 dcn direct ops pairmlp; do $OBJCOPY --redefine-syms=$B/redefine.map $B/o3/$f.o $B/o3/$f.twin.o; TW="$TW $B/o3/$f.twin.o"; done
$CXX -shared -pthread -o $B/libdeft_emu.so $B/o2/igemm.o $B/o2/igemm3.o $B/o2/dcn.o $B/o2/direct.o $B/o2/ops.o $B/o2/pairmlp.o $B/o2/assoc.o $TW
echo built tests/hipemu/_build/libdeft_emu.so
