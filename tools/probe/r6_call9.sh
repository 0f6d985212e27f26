# round 6, call 9: bisection of the round-4 packed-fp32 fault with wait states at two places of igemm.hip MODE_DCN (built with v_pk_*_f32 allowed)
for l in pkf32 pkf32_a pkf32_b pkf32_ab hip; do
  echo "== $l"
  DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$l.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "test_launches_bit_exact_beside_another_kernel" 2>&1 | grep -E "passed|failed|AssertionError: launch" | head -3
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "beside_another_kernel or teardown" 2>&1 | tail -2
