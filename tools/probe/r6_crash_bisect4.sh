#!/bin/bash
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -s -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(grep -v '^$' gpurun_out/r6x/$name.log | tail -1 | cut -c1-80)"; }
DEFT_DEBUG_CAPTURE=1 run dbg $t8 $t9 $t12
grep "capture_graph" gpurun_out/r6x/dbg.log | tail -12
DEFT_DATAFLOW=1 run nodataflow $t8 $t9 $t12
