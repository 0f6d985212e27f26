#!/bin/bash
# round 6: is the hipGraph-replay segfault a matter of WHEN the garbage of earlier tests (Detectors, slots, graphs in reference cycles) is collected?
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/r6x/$name.log | cut -c1-80)"; }
run base $t8 $t9 $t12
DEFT_TEST_GC=each run gc_each $t8 $t9 $t12
DEFT_TEST_GC=off run gc_off $t8 $t9 $t12
AMD_SERIALIZE_KERNEL=3 run serialize $t8 $t9 $t12
HIP_LAUNCH_BLOCKING=1 run blocking $t8 $t9 $t12
base2() { run base2 $t8 $t9 $t12; }; base2
