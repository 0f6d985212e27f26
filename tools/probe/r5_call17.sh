#!/bin/bash
# round 5, GPU call 17: ArrayTracker.announce (the next frame's embedding / affinity chain queued while this frame is associated) in the bench's own regime,
# against DEFT_BEGIN_AHEAD=0; the fused-run / tracker tests on the device first
mkdir -p gpurun_out/r5q
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "track or fused_run" > gpurun_out/r5q/tests.log 2>&1
tail -2 gpurun_out/r5q/tests.log
for ba in 1 0 1; do
  DEFT_BEGIN_AHEAD=$ba timeout 200 python tools/probe/r5_e2e_ab.py B 2>/dev/null | grep '^{' | tee -a gpurun_out/r5q/e2e_ab.log
done
DEFT_BEGIN_AHEAD=1 timeout 200 python tools/probe/r5_e2e_ab.py D 2>/dev/null | grep '^{' | tee -a gpurun_out/r5q/e2e_ab.log
