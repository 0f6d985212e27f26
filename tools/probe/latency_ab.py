"""A/B of the one-frame-per-step (latency) mode under engine switches given as NAME=value arguments, e.g.
    python tools/probe/latency_ab.py DCN_PATCH_MIN_TILES=64 P3_MIN_TILES=256
Prints ms per frame of detect + embed + 100x500 affinity under hipGraph replay (the bench's latency_mode)."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from deft_amd import engine, hiplib, synth  # noqa: E402
from deft_amd.pipeline import FramePipeline, HipCompute  # noqa: E402

overlap = False
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "OVERLAP":                       # HipCompute(overlap=...): detection of step k+1 on a side stream next to step k's affinity chain
        overlap = bool(int(v)); continue
    setattr(engine, k, type(getattr(engine, k))(float(v)) if not isinstance(getattr(engine, k), bool) else bool(int(v)))
lib = hiplib.get_lib()
sd = synth.synth_state_dict("mot")
H, W = 608, 1088
dev = torch.device("cuda")
comp = HipCompute(sd, 1, H, W, "mot", K=100, device=dev, lib=lib, streams=1, ndet=100, overlap=overlap)
pipe = FramePipeline(comp, 1, 100, comp.D, history=5, device=dev, exchange=False)
x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
comp.capture(x)
for _ in range(10):
    pipe.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
for _ in range(n):
    pipe.step(x)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
kinds = {}
for e, _, d in comp.plan._gemms:
    if e == "deft_dcn_v2_nhwc":
        kinds["dcn patch" if d.p3_kernel == 2 else "dcn igemm S=%d" % max(1, d.splitk)] = kinds.get("dcn patch" if d.p3_kernel == 2 else "dcn igemm S=%d" % max(1, d.splitk), 0) + 1
print("%s -> %.3f ms/frame  %s" % (" ".join(sys.argv[1:]) or "(defaults)", ms, kinds))
