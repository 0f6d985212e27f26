#!/bin/bash
# round 5, GPU call 6: the default bench line as the driver runs it (every extra, side configs, the bf16x3 build next to it, cpu baseline)
mkdir -p gpurun_out/r5f
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5f/bench_default.json 2> gpurun_out/r5f/bench_default.err
tail -5 gpurun_out/r5f/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5f/bench_default.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms")
print(json.dumps(d["config"], indent=1))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "peak", "frac", "serialized_frac", "frac_of_bf16x3_ceiling")}, d["roofline"]["dominant_kernel"])
print("cpu", d["cpu_baseline"])
print("keys", sorted(d))
PY
