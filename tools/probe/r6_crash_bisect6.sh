#!/bin/bash
# is the hipGraph-replay segfault an uninitialised read (heap-state dependent)?  glibc's MALLOC_PERTURB_ fills every allocated / freed block with a byte pattern
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
run() { name=$1; shift; timeout 600 python -X faulthandler -m pytest -x -q -p no:cacheprovider "$@" > gpurun_out/r6x/$name.log 2>&1; echo "$name rc=$? $(grep -v '^$' gpurun_out/r6x/$name.log | tail -1 | cut -c1-80)"; }
MALLOC_PERTURB_=165 run perturb_b $t12
MALLOC_PERTURB_=165 run perturb_u8_b $t9 $t12
MALLOC_PERTURB_=165 timeout 300 python -X faulthandler tools/probe/graph_replay_count.py 6 > gpurun_out/r6x/perturb_count.log 2>&1; echo "perturb count rc=$? $(tail -1 gpurun_out/r6x/perturb_count.log | cut -c1-100)"
MALLOC_PERTURB_=165 DEFT_DATAFLOW=1 run perturb_nodataflow $t12
