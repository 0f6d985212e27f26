#!/bin/bash
# round 6: the order-dependent segfault in hipGraph replay, second cut: subsets of the sequence that reproduces it
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t1="$T::test_embed_fused"; t2="$T::test_lstm[nuscenes]"; t3="$T::test_motion_step[nuscenes]"; t4="$T::test_forward_embed_affinity_golden[nuscenes_96x128-nuscenes-96-128]"
t5="$T::test_topk_seed_sweep_both_arithmetics[nuscenes-448-800-16]"; t6="$T::test_full_size_other_configs[nuscenes-448-800]"; t7="$T::test_seam_model_afe_decode[nuscenes]"
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/r6x/$name.log | cut -c1-80)"; }
run all $t1 $t2 $t3 $t4 $t5 $t6 $t7 $t8 $t9 $t12
run c1 $t5 $t6 $t7 $t8 $t9 $t12
run c2 $t8 $t9 $t12
run c3 $t5 $t6 $t9 $t12
run c4 $t1 $t2 $t3 $t4 $t9 $t12
run c5 $t5 $t6 $t7 $t8 $t12
run c6 $t6 $t9 $t12
run c7 $t5 $t9 $t12
