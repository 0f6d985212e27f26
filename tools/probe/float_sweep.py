"""the FLOAT errors of the device path against the oracle (embedding, bbox, score, heat-map logit, affinity) swept over input
seeds ON THE TIMED PLANS (32 frames per step as two 16-frame sub-batch plans on two HIP streams), for BOTH arithmetics of the library on one oracle
pass.  Run on the GPU box:   python tools/probe/float_sweep.py CONFIG NSEEDS   ->  gpurun_out/float_sweep_<CONFIG>.json (copy to profiles/)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import deft_oracle as O  # noqa: E402
from deft_amd import hiplib  # noqa: E402

name, nseeds = sys.argv[1], int(sys.argv[2])
cfg = bench.CONFIGS[name]
H, W, ds = cfg["H"], cfg["W"], cfg["dataset"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = hiplib.get_lib()
libs = {"fp16x2": lib, "bf16x3": lib.twin()}
B = 32
torch.set_num_threads(32)
frames = [torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(2000 + s)) for s in range(nseeds)]
res = {k: [] for k in libs}
t0 = time.time()
wls = {k: bench.build_workload(cfg, B, 2, dev, l, 0) for k, l in libs.items()}
for s0 in range(0, nseeds, B):
    ids = [(s0 + b) % nseeds for b in range(B)]
    images = torch.cat([frames[i] for i in ids], 0).to(dev)
    snaps = {}
    for k, wl in wls.items():
        wl["step"](images)
        snaps[k] = bench.gate_snapshot(wl, wl["step"](images))          # twice: the history ring in steady state on these frames
    for b, i in enumerate(ids):
        if i < s0:
            continue                                                    # (a wrapped-around seed: checked in its own batch)
        with torch.no_grad():
            out, maps = O.dlaseg_forward(frames[i], wls["fp16x2"]["sd"], ds)
            od = O.generic_decode(O.sigmoid_output(out), K=bench.KDET)
        for k in libs:
            r, errs = bench._gate_frame(cfg, wls[k]["sd"], images, snaps[k], b, oracle=(out, maps, od))
            res[k].append(dict(errs, seed=2000 + i, equal=r["topk_ordered_equal"], tie_class=r["differences_within_the_oracles_tie_class"], margin=r["oracle_margin"]))
        print(name, "seed", 2000 + i, {k: "%.2e" % res[k][-1]["embedding"] for k in libs}, "%.0f s" % (time.time() - t0), flush=True)
rep = {"config": name, "seeds": nseeds, "plans": "32 frames per step as 2 sub-batch plans of 16 on 2 HIP streams (the timed plans' configuration)"}
for k in libs:
    rep[k] = {"max": {q: max(e[q] for e in res[k]) for q in ("embedding", "bbox", "score", "hm_logit", "affinity")},
              "median": {q: sorted(e[q] for e in res[k])[len(res[k]) // 2] for q in ("embedding", "bbox", "score", "hm_logit", "affinity")},
              "frames_ordered_equal": sum(e["equal"] for e in res[k]), "frames_outside_tie_class": sum(1 for e in res[k] if not e["equal"] and not e["tie_class"]),
              "seeds_not_ordered_equal": [e["seed"] for e in res[k] if not e["equal"]],
              "seeds_outside_tie_class": [e["seed"] for e in res[k] if not e["equal"] and not e["tie_class"]],
              "per_seed_embedding": [round(e["embedding"], 7) for e in res[k]]}
print(json.dumps(rep))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "float_sweep_%s.json" % name), "w"), indent=1)
