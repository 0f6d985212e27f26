#!/bin/bash
# round 6: end to end after the host-side trims (ResultList arrays, native node selection / gather table) -- B, D, E, two repetitions; then the tests of the loop
mkdir -p gpurun_out/r6q
O=gpurun_out/r6q
for rep in 1 2; do
for c in B D E; do
  timeout 400 python bench.py --e2e-only $c --e2e-frames 200 > $O/e2e2_${c}_$rep.json 2> $O/e2e2_${c}_$rep.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/e2e2_${c}_$rep.json") if l.startswith("{")][-1])
print("$c rep $rep:", d["value"], "frames/s", d["runs_ms_per_frame"], "one-frame", d["one_frame_lookahead"]["ms_per_frame"], "serial", d["serial"]["ms_per_frame"], "eight", (d.get("eight_frames_per_pass") or {}).get("value"), d["stage_ms"])
PY
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or lookahead or prefetch or tracker or stream or e2e or teardown or abi" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
