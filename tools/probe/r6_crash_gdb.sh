#!/bin/bash
# native backtrace + the faulting instruction of the hipGraph-replay segfault
mkdir -p gpurun_out/r6x
T=tests/test_gpu_parity.py
t8="$T::test_nuscenes_run_replays_reference_trace"; t9="$T::test_fused_detector_run_on_uint8_frames"; t12="$T::test_fused_run_with_lookahead"
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop" -ex run -ex "bt 6" -ex "info registers" -ex "x/70i \$rip-200" --args python -m pytest -x -q -p no:cacheprovider $t8 $t9 $t12 > gpurun_out/r6x/gdb2.log 2>&1
grep -n "SIGSEGV" -A140 gpurun_out/r6x/gdb2.log | grep -v "^--" | head -150
