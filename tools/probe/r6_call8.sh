# round 6, call 8: the whole GPU suite on the tree; the round-4 packed-fp32 fault's precondition (igemm.hip with v_pk_*_f32 allowed) beside a co-resident
# kernel; the readiness script on one GPU; the gate's tie class over 64 seeds
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r6c8_gpu_tests.log 2>&1; tail -4 gpurun_out/r6c8_gpu_tests.log
echo "== packed fp32 allowed in igemm.hip (libdeft_pkf32.so): co-residency tests"
DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_pkf32.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "beside_another_kernel" > gpurun_out/r6c8_pkf32.log 2>&1; tail -6 gpurun_out/r6c8_pkf32.log
echo "== scale_check.sh 1 2"
bash tools/scale_check.sh 1 2 2>&1 | tail -12
timeout 1500 python tools/probe/float_sweep.py B 64 > gpurun_out/r6c8_sweep_B.log 2>&1; tail -1 gpurun_out/r6c8_sweep_B.log | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
for k in ('fp16x2', 'bf16x3'):
    print(k, j[k]['max'], 'equal', j[k]['frames_ordered_equal'], 'not equal', j[k]['seeds_not_ordered_equal'], 'outside tie class', j[k]['seeds_outside_tie_class'])"
