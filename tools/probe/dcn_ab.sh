# A/B of builds of the DCN patch kernel on the layer shapes of config B: libdeft_<variant>.so (tools/build_variant.sh) vs libdeft_hip.so
# usage: bash tools/probe/dcn_ab.sh [variant ...]   (default: base)
VARS=${@:-base}
for rep in 1 2; do
for lib in $VARS hip; do
  echo "== $lib"
  for shape in "152 272 64 64 16 64" "76 136 128 64 16 64" "38 68 256 64 16 64" "76 136 128 128 16 128" "38 68 256 256 16 128"; do
    DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$lib.so OFFSET_SIGMA=${OFFSET_SIGMA:-1.5} timeout 60 python tools/probe/dcnp_one.py $shape 20 2>&1 | tail -1
  done
done
done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "dcn" 2>&1 | tail -3
