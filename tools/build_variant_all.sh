#!/bin/bash
# A/B build of the WHOLE library with extra compiler flags:  tools/build_variant_all.sh <name> [flags...]  -> deft_amd/lib/libdeft_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
python -m deft_amd.build > /dev/null
OBJS=""
for f in igemm.hip igemm3.hip dcn.hip direct.hip ops.hip pairmlp.hip; do
    EXTRA=""; [ "$f" = "dcn.hip" ] && EXTRA="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA "$@" -c deft_amd/csrc/$f -o /tmp/va_${NAME}_$f.o 2> >(grep -v "packed-fp32-ops" >&2) &
    OBJS="$OBJS /tmp/va_${NAME}_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o deft_amd/lib/libdeft_$NAME.so $OBJS deft_amd/lib/obj/assoc.hip.o
echo deft_amd/lib/libdeft_$NAME.so
