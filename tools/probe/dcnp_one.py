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
if os.environ.get("DCNP_TIMING"):
    nwg = B * -(-H // 8) * -(-W // 16) * -(-Co // (patch or 64))
    tbuf = torch.zeros(nwg * 4 * 12, dtype=torch.int64, device="cuda")
    d.ws = tbuf.data_ptr()
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

if os.environ.get("DCNP_TIMING"):
    nwg_ = tbuf.numel() // 48
    t = tbuf.cpu()[:nwg_ * 32].view(-1, 4, 8).double()
    ts = tbuf.cpu()[nwg_ * 32:].view(-1, 4, 4).double()
    nst = Ci // 16 * 9
    for wv in range(4):
        print("   wave %d per-step ticks: wait+barrier %.0f | gather+mfma+blend %.0f | dma issue %.0f | read_b (issue + landed) %.0f" % ((wv,) + tuple(float(ts[:, wv, i].mean()) / nst for i in range(4))))
    names = ["dma-issue", "records", "first wait+gather", "loop", "sync+stage", "store issue", "store drain"]
    dt = t[:, :, 1:] - t[:, :, :-1]
    print("per-wave phase cycles (mean over %d workgroups x 4 waves; s_memtime ticks):" % t.shape[0])
    for i, n in enumerate(names):
        print("   %-20s mean %9.0f   p10 %9.0f  p90 %9.0f" % (n, float(dt[:, :, i].mean()), float(dt[:, :, i].flatten().kthvalue(max(1, int(0.1 * dt[:, :, i].numel()))).values), float(dt[:, :, i].flatten().kthvalue(int(0.9 * dt[:, :, i].numel())).values)))
    life = t[:, :, 7] - t[:, :, 0]
    print("   %-20s mean %9.0f" % ("total", float(life.mean())))
    span = float(t[:, :, 7].max() - t[:, :, 0].min())
    print("   kernel span %.0f ticks; sum of wave lives / span = %.2f waves resident" % (span, float(life.sum()) / span))
