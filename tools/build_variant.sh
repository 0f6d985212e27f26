#!/bin/bash
# A/B builds of one HIP source:  tools/build_variant.sh <name> <file.hip> [flags...]  -> deft_amd/lib/libdeft_<name>.so (select it with DEFT_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
python -m deft_amd.build > /dev/null
EXTRA=""; [ "$SRC" = "dcn.hip" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA "$@" -c deft_amd/csrc/$SRC -o /tmp/variant_$NAME.o
OBJS=""
for f in igemm.hip igemm3.hip dcn.hip direct.hip ops.hip pairmlp.hip assoc.hip; do
    if [ "$f" = "$SRC" ]; then OBJS="$OBJS /tmp/variant_$NAME.o"; else OBJS="$OBJS deft_amd/lib/obj/$f.o"; fi
done
# (the three-bf16-piece twins of the product library ride along unchanged: deft_amd/build.py)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o deft_amd/lib/libdeft_$NAME.so $OBJS $(ls deft_amd/lib/obj_twin/*.o)
echo deft_amd/lib/libdeft_$NAME.so
