#!/bin/bash
# round 5, GPU call 19: the closing state once more on the device -- smoke(), the tracker / fused-run tests (graph capture with the GC held off)
mkdir -p gpurun_out/r5s
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5s/smoke.log 2>&1; tail -1 gpurun_out/r5s/smoke.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "track or fused_run or latency or graph" > gpurun_out/r5s/tests.log 2>&1
grep -v "^Extension modules" gpurun_out/r5s/tests.log | tail -3
