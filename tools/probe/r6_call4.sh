# round 6, call 4: both arithmetics in one library on the hardware; the fixed floor of the DCN launch; the gate with the steady-state decidable stream
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "out_of_range or twin_arithmetic or producer_consumer or test_dcn_patch" > gpurun_out/r6c4_tests.log 2>&1; tail -3 gpurun_out/r6c4_tests.log
export OFFSET_SIGMA=1.5
for lib in pc_floor hip; do
  for shape in "152 272 64 64 16 64" "76 136 128 64 16 64"; do
    echo -n "$lib PC=1: "; DEFT_DCN_PC=1 DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$lib.so timeout 120 python tools/probe/dcnp_one.py $shape 20 2>&1 | tail -1
  done
done
echo -n "one-role: "; timeout 120 python tools/probe/dcnp_one.py 152 272 64 64 16 64 20 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r6c4_bench.log 2>&1; tail -c 5000 gpurun_out/r6c4_bench.log
