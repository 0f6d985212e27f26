#!/bin/bash
# round 6: ArrayTracker.prepare (affinity blocks one frame further ahead) -- end-to-end A/B inside one call, then the GPU tests that touch the loop
mkdir -p gpurun_out/r6q
O=gpurun_out/r6q
for rep in 1 2; do
for pa in 1 0; do
  DEFT_PREPARE_AHEAD=$pa timeout 400 python bench.py --e2e-only B --e2e-frames 200 > $O/e2e_B_pa${pa}_$rep.json 2> $O/e2e_B_pa${pa}_$rep.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/e2e_B_pa${pa}_$rep.json") if l.startswith("{")][-1])
print("prepare_ahead=$pa rep $rep B:", d["value"], "frames/s", d["runs_ms_per_frame"], "one-frame", d["one_frame_lookahead"]["ms_per_frame"], "serial", d["serial"]["ms_per_frame"], "eight", (d.get("eight_frames_per_pass") or {}).get("value"), d["stage_ms"])
PY
done
done
for c in D; do
for pa in 1 0; do
  DEFT_PREPARE_AHEAD=$pa timeout 400 python bench.py --e2e-only $c --e2e-frames 100 > $O/e2e_${c}_pa$pa.json 2> $O/e2e_${c}_pa$pa.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/e2e_${c}_pa$pa.json") if l.startswith("{")][-1])
print("prepare_ahead=$pa $c:", d["value"], "frames/s", d["runs_ms_per_frame"], "eight", (d.get("eight_frames_per_pass") or {}).get("value"))
PY
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or lookahead or prefetch or tracker or stream or e2e or teardown" > $O/tests.log 2>&1; tail -3 $O/tests.log
