#!/bin/bash
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t5="$T::test_topk_seed_sweep_both_arithmetics[nuscenes-448-800-16]"; t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(grep -v '^$' gpurun_out/r6x/$name.log | tail -1 | cut -c1-80)"; }
run fix_c2 $t8 $t9 $t12
run fix_c7 $t5 $t9 $t12
DEFT_REPLAY_STREAM=0 run nofix_c2 $t8 $t9 $t12
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q -k "nuscenes or ddd or abi or fused" > gpurun_out/r6x/fix_sel.log 2>&1; echo "selection rc=$? $(tail -1 gpurun_out/r6x/fix_sel.log | cut -c1-80)"
