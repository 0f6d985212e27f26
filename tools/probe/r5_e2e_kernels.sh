#!/bin/bash
# round 5, GPU call 20: rocprofv3 --kernel-trace --stats of the end-to-end loop of config B (four frames per lookahead pass, 48 warm-up + 100 timed frames,
# nothing else in the process): how much device time a tracked frame costs, by kernel -- the pass against the tracker's affinity chain
ROOT=$PWD
mkdir -p gpurun_out/r5t
cd /tmp && export TMPDIR=/tmp
DEFT_E2E_REPORTED_MODE_ONLY=1 timeout 170 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r5t/stats -o e2eB --output-format csv -- \
    python $ROOT/bench.py --e2e-only B --e2e-frames 100 > $ROOT/gpurun_out/r5t/run.log 2>&1
cd $ROOT
grep '^{' gpurun_out/r5t/run.log | cut -c1-400
f=$(find gpurun_out/r5t/stats -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5t/e2eB_kernel_stats.csv
find gpurun_out/r5t/stats -name '*kernel_trace.csv' -delete
head -25 gpurun_out/r5t/e2eB_kernel_stats.csv | cut -c1-200
