"""profiles/<tag>_layers.md from the per-launch HIP-event records bench.py leaves in gpurun_out/<tag>/bench_ops.json
(one profiled step, launches serialised on one stream).  Usage: python tools/layers_table.py r1"""
import collections
import json
import os
import sys

tag = sys.argv[1]
d = json.load(open(os.path.join("gpurun_out", tag, "bench_ops.json")))
rows = collections.OrderedDict()
for c in d["calls"]:
    name, fl, ms, info = c[:4]
    k = (name, info)
    r = rows.setdefault(k, [0, 0.0, 0.0, c[4] if len(c) > 4 else 157.3])
    r[0] += 1; r[1] += ms; r[2] += fl
tot = sum(r[1] for r in rows.values())
frames = 64
try:                                     # frames per step from the bench line of the same run (tools/prof.sh keeps it beside the records)
    line = [l for l in open(os.path.join("gpurun_out", tag, "bench.log")) if l.startswith("{")][-1]
    frames = int(json.loads(line)["config"]["frames_per_step_per_gpu"])
except Exception:
    pass
out = ["# Per-layer timing of one profiled step (`profiles/%s`, %d frames = 2 sub-batches of %d, launches serialised on one stream)\n" % (tag, frames, frames // 2),
       "HIP-event bracket per launch (includes ≈6 µs of bracket overhead); FLOPs are algorithmic (no padding). Total %.2f ms.\n" % tot,
       "Ceiling = what the launch's matrix instructions can do: 416.7 TFLOP/s fp32-equivalent for the split-bf16 launches (2500 / 6), "
       "157.3 for v_mfma_f32_32x32x2_f32; 833.3 for the two-fp16-piece launches (2500 / 3).\n",
       "| entry | shape | launches | ms | % | TFLOP/s | ceiling | frac |", "|---|---|---|---|---|---|---|---|"]
for (name, info), (n, ms, fl, ceil) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tf = fl / ms / 1e9 if fl else None
    out.append("| `%s` | %s | %d | %.3f | %.1f | %s | %s | %s |" % (name, info or "-", n, ms, 100 * ms / tot,
                                                                 "%.1f" % tf if tf else "-", "%.1f" % ceil if tf else "-", "%.2f" % (tf / ceil) if tf else "-"))
open(os.path.join("profiles", tag + "_layers.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
