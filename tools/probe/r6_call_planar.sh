#!/bin/bash
# round 6: the image layer reading the [N,3,H,W] planes itself (no layout pass): targeted GPU tests, then the bench without its side legs
mkdir -p gpurun_out/r6p
O=gpurun_out/r6p
timeout 1200 python -m pytest tests -m gpu -x -q -k "direct or frame_pipeline or detector or u8 or plan or process or golden or flip" > $O/tests.log 2>&1
tail -4 $O/tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err || tail -5 $O/bench_$i.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r6p/bench_$i.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", d["dtype"], d["config"]["parity"]["pass"])
PY
done
