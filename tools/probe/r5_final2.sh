#!/bin/bash
# round 5, final GPU call after the tracker's native host half (deft_associate_2d / _ddd, deft_kf_*, ArrayTracker.begin): the full -m gpu suite on the
# product library and the default bench line as the driver runs it.  The kernels are the ones of r5_final.sh (only csrc/assoc.hip's host functions
# changed): the rocprofv3 passes of that call stand.
mkdir -p gpurun_out/r5y
O=gpurun_out/r5y
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
cp gpurun_out/bench_ops.json $O/bench_ops.json
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5y/bench_full.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", json.dumps(d["config"]["parity"]), json.dumps(d["config"]["side"]))
print("e2e B", json.dumps({k: d["end_to_end"][k] for k in ("ms_per_frame", "value", "stage_ms")}))
for n in ("D", "E"):
    e = d["configs"][n]["end_to_end"]; print("e2e", n, json.dumps({k: e[k] for k in ("ms_per_frame", "value", "stage_ms")}))
print("C tracked", json.dumps(d["config_C"]["tracked"])[:300])
PY
