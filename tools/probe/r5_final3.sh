#!/bin/bash
# round 5, last GPU call: the default bench line as the driver runs it, with the end-to-end legs in processes of their own (bench.py --e2e-only).
# deft_amd/ and tests/ are those of r5_final2.sh's call (183 device tests green there); the kernels those of r5_final.sh (rocprofv3 passes).
mkdir -p gpurun_out/r5w
O=gpurun_out/r5w
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
cp gpurun_out/bench_ops.json $O/bench_ops.json
tail -5 $O/bench_full.err | cut -c1-300
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5w/bench_full.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", json.dumps(d["config"]["parity"]), json.dumps(d["config"]["side"]))
e = d["end_to_end"]; print("e2e B", json.dumps({k: e.get(k) for k in ("ms_per_frame", "value", "runs_ms_per_frame", "stage_ms", "process", "eight_frames_per_pass")}), e["one_frame_lookahead"]["ms_per_frame"], e["serial"]["ms_per_frame"], e["serial"]["stage_ms"]["track"])
for n in ("D", "E"):
    e = d["configs"][n]["end_to_end"]; print("e2e", n, json.dumps({k: e.get(k) for k in ("ms_per_frame", "value", "runs_ms_per_frame", "stage_ms", "process")}))
print("C tracked", json.dumps(d["config_C"]["tracked"])[:200])
PY
