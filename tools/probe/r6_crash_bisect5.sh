#!/bin/bash
mkdir -p gpurun_out/r6x; rm -f gpurun_out/r6x/sched_*.json
T=tests/test_gpu_parity.py
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
DEFT_DEBUG_CAPTURE=1 timeout 600 python -X faulthandler -m pytest -x -q -s -p no:cacheprovider $t8 $t9 $t12 > gpurun_out/r6x/dbg5.log 2>&1; echo "crash run rc=$?"
mkdir -p gpurun_out/r6x/ok; mv gpurun_out/r6x/sched_*.json gpurun_out/r6x/ok/ 2>/dev/null; mkdir -p gpurun_out/r6x/bad; mv gpurun_out/r6x/ok/* gpurun_out/r6x/bad/
DEFT_DEBUG_CAPTURE=1 timeout 600 python -X faulthandler -m pytest -x -q -s -p no:cacheprovider $t9 $t12 > gpurun_out/r6x/dbg5ok.log 2>&1; echo "ok run rc=$?"
mv gpurun_out/r6x/sched_*.json gpurun_out/r6x/ok/
ls gpurun_out/r6x/bad gpurun_out/r6x/ok
