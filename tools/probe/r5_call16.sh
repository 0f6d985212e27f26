#!/bin/bash
# round 5, GPU call 16: end to end on config B in the bench's own regime (100 timed frames, 50 stored frames) with and without ArrayTracker.begin ahead
mkdir -p gpurun_out/r5p
for ba in 0 1; do
  DEFT_BEGIN_AHEAD=$ba timeout 200 python tools/probe/r5_e2e_ab.py B 2>/dev/null | grep '^{' | tee -a gpurun_out/r5p/e2e_ab.log
done
DEFT_BEGIN_AHEAD=0 timeout 200 python tools/probe/r5_e2e_ab.py E 2>/dev/null | grep '^{' | tee -a gpurun_out/r5p/e2e_ab.log
