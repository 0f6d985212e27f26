# A/B of two builds of the DCN patch kernel on the layer shapes of config B: libdeft_base.so (tools/build_variant.sh) vs libdeft_hip.so
for lib in base hip base hip base hip; do
  echo "== $lib"
  for shape in "152 272 64 64 16 64" "76 136 128 64 16 64" "76 136 128 128 16 128" "38 68 256 256 16 128"; do
    DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$lib.so OFFSET_SIGMA=${OFFSET_SIGMA:-1.5} timeout 60 python tools/probe/dcnp_one.py $shape 20 2>&1 | tail -1
  done
done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "dcn" 2>&1 | tail -3
