#!/bin/bash
# round 6, closing GPU call: the default bench line as the driver runs it, then the rocprofv3 passes (stats, SQ, FETCH, WRITE) of the same build
mkdir -p gpurun_out/r6z
O=gpurun_out/r6z
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
cp gpurun_out/bench_ops.json $O/bench_ops.json
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6z/bench_full.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", d["dtype"]); print(json.dumps(d["config"]["parity"])[:600]); print(json.dumps(d["config"]["side"])); print(json.dumps(d["cpu_baseline"]))
print(json.dumps({k: d["roofline"][k] for k in ("achieved", "peak", "frac", "serialized_frac", "dominant_kernel", "traffic_bytes_per_step")}))
PY
bash tools/prof.sh r6 > $O/prof.log 2>&1
tail -3 $O/prof.log
