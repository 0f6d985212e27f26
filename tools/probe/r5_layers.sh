#!/bin/bash
# per-layer HIP-event table of the product build (one profiled step, launches serialised on one stream)
mkdir -p gpurun_out/r5
timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5/bench_noextras.json 2>/dev/null
cp gpurun_out/bench_ops.json gpurun_out/r5/bench_ops.json
python tools/layers_table.py r5 | head -12
