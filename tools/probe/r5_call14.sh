#!/bin/bash
# round 5, GPU call 14: the tracker's host half through deft_associate_2d / deft_kf_* -- tracker tests on the device, then where update()'s time goes now
mkdir -p gpurun_out/r5n
timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -k "track or fused_run or association or similarity or motion" > gpurun_out/r5n/tests.log 2>&1
tail -3 gpurun_out/r5n/tests.log
timeout 200 python tools/probe/r5_e2e_profile.py B > gpurun_out/r5n/e2e_B.log 2>&1
head -45 gpurun_out/r5n/e2e_B.log | cut -c1-180
timeout 200 python tools/probe/r5_e2e_profile.py D > gpurun_out/r5n/e2e_D.log 2>&1
sed -n 2,3p gpurun_out/r5n/e2e_D.log | cut -c1-400
