#!/bin/bash
# round 5, GPU call 15: ArrayTracker.begin -- the next frame's device half queued behind update(k) (DEFT_BEGIN_AHEAD=1, default) against every frame's
# device half inside its own update() (=0); tracker / fused-run tests on the device first
mkdir -p gpurun_out/r5o
timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -k "track or fused_run" > gpurun_out/r5o/tests.log 2>&1
tail -3 gpurun_out/r5o/tests.log
for ba in 0 1; do
  DEFT_BEGIN_AHEAD=$ba timeout 200 python tools/probe/r5_e2e_profile.py B > gpurun_out/r5o/e2e_B_begin$ba.log 2>&1
  echo "begin_ahead=$ba"; sed -n 2,3p gpurun_out/r5o/e2e_B_begin$ba.log | cut -c1-330
  grep -m3 "synchronize\|hiplib.py\|(update)" gpurun_out/r5o/e2e_B_begin$ba.log | cut -c1-150
done
DEFT_BEGIN_AHEAD=1 timeout 200 python tools/probe/r5_e2e_profile.py D > gpurun_out/r5o/e2e_D_begin1.log 2>&1
sed -n 2,3p gpurun_out/r5o/e2e_D_begin1.log | cut -c1-330
