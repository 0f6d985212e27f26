#!/bin/bash
# round 6: the order-dependent segfault in hipGraph replay (test_fused_run_with_lookahead after the nuScenes tests): which predecessor does it need?
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
B="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(grep -c PASSED gpurun_out/r6x/$name.log) $(tail -1 gpurun_out/r6x/$name.log | cut -c1-80)"; }
run b_alone $B
run u8_b "$T::test_fused_detector_run_on_uint8_frames" $B
run nusc_trace_b "$T::test_nuscenes_run_replays_reference_trace" $B
run fullsize_b "$T::test_full_size_other_configs[nuscenes-448-800]" $B
run sweep_b "$T::test_topk_seed_sweep_both_arithmetics[nuscenes-448-800-16]" $B
run seam_b "$T::test_seam_model_afe_decode[nuscenes]" $B
run golden_b "$T::test_forward_embed_affinity_golden[nuscenes_96x128-nuscenes-96-128]" $B
run lstm_b "$T::test_lstm[nuscenes]" "$T::test_motion_step[nuscenes]" "$T::test_embed_fused" $B
