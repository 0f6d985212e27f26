#!/bin/bash
# round 5, GPU call 13: frames per step / HIP streams sweep with the two-piece kernels (round 3's sweep chose 32 frames on 2 streams)
mkdir -p gpurun_out/r5m
for cfg in "32 2" "32 4" "48 3" "64 2" "64 4" "48 2" "32 2"; do
  set -- $cfg
  timeout 300 python bench.py --batch $1 --streams $2 --steps 20 --warmup 4 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $1 streams $2:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
"
done > gpurun_out/r5m/batch_streams.log 2>&1
cat gpurun_out/r5m/batch_streams.log
