#!/bin/bash
# does the runtime's "parallel stream equals the launch stream" test follow the hardware-queue mapping?  The two-branch reproducer under GPU_MAX_HW_QUEUES
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(grep -v '^$' gpurun_out/r6x/$name.log | tail -1 | cut -c1-80)"; }
export DEFT_DATAFLOW=2
run hwq_default $t8 $t9 $t12
for q in 1 2 8 16; do GPU_MAX_HW_QUEUES=$q run hwq_$q $t8 $t9 $t12; done
