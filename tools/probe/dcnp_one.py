"""Profiling aid (GPU): ONE DCN main-contraction launch shape, repeated -- `python tools/probe/dcnp_one.py H W Cin Cout [batch] [patch: 0|64|128] [reps]`.
Run under rocprofv3 --pmc to read the counters of dcn_patch_kernel / igemm_kernel<MODE_DCN> alone."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from deft_amd import engine, hiplib  # noqa: E402

H, W, Ci, Co = (int(a) for a in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
patch = int(sys.argv[6]) if len(sys.argv) > 6 else 64
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
lib = hiplib.get_lib()
g = torch.Generator().manual_seed(0)
sd = {"d.conv.weight": torch.randn(Co, Ci, 3, 3, generator=g) * 0.05, "d.conv.bias": torch.zeros(Co),
      "d.conv.conv_offset_mask.weight": torch.randn(27, Ci, 3, 3, generator=g) * 0.01,
      "d.conv.conv_offset_mask.bias": torch.randn(27, generator=g) * float(os.environ.get("OFFSET_SIGMA", "0.5")),
      "d.actf.0.weight": torch.ones(Co), "d.actf.0.bias": torch.zeros(Co),
      "d.actf.0.running_mean": torch.zeros(Co), "d.actf.0.running_var": torch.ones(Co)}
engine.DCN_PATCH, engine.DCN_PATCH_MIN_TILES, engine.DCN_PATCH_WASTE = bool(patch), 0, 1e9
plan = engine.DlaSegPlan.__new__(engine.DlaSegPlan)
engine._Plan.__init__(plan, "cuda", lib)
plan.sd = sd; plan._wcache = {}
xv = plan.alloc(B, H, W, Ci); xv.buf.normal_()
plan._deform("d", xv)
d = plan._keep[-1]
if patch:
    d.tile = patch
off_op, dcn_op = plan.ops[-2], plan.ops[-1]
off_op[2](); torch.cuda.synchronize()
plan.ops = [dcn_op]
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    plan.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("dcn %dx%d %d->%d B=%d patch=%d: %.3f ms %.1f TF/s" % (H, W, Ci, Co, B, patch, ms, dcn_op[3] / ms / 1e9))
