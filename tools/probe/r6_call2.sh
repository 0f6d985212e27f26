# round 6, call 2: the producer / consumer DCN kernel on the hardware -- parity first, then the A/B against the one-role kernel, counters, the new gate
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dcn" > gpurun_out/r6c2_tests.log 2>&1; tail -3 gpurun_out/r6c2_tests.log
export OFFSET_SIGMA=1.5
for rep in 1 2; do
for pcv in 1 0; do
  echo "== DEFT_DCN_PC=$pcv"
  for shape in "152 272 64 64 16 64" "76 136 128 64 16 64" "38 68 256 64 16 64"; do
    DEFT_DCN_PC=$pcv timeout 120 python tools/probe/dcnp_one.py $shape 20 2>&1 | tail -1
  done
done
done > gpurun_out/r6c2_ab.log 2>&1
cat gpurun_out/r6c2_ab.log
bash tools/pmc.sh r6c2 'dcn_pc' -- python tools/probe/dcnp_one.py 152 272 64 64 16 64 5 > gpurun_out/r6c2_pmc.log 2>&1
cat gpurun_out/r6c2/pmc*.txt
timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r6c2_bench.log 2>&1; tail -c 6000 gpurun_out/r6c2_bench.log
