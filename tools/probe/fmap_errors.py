import sys, os
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/oracle']
import torch, deft_oracle as O, parity_checks as pc
from deft_amd import engine, hiplib
ds, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sd = O.synth_state_dict(ds)
torch.set_num_threads(32)
x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(1000))
with torch.no_grad():
    out, maps = O.dlaseg_forward(x, sd, ds)
for prec in (0, 1):
    engine.PREC = prec
    plan = engine.DlaSegPlan(sd, 1, H, W, ds, K=100, device="cuda", lib=hiplib.get_lib())
    plan.forward(x.cuda()); torch.cuda.synchronize()
    errs = [pc.maxabs(fm.to_nchw().cpu(), m) / max(1.0, float(m.abs().max())) for fm, m in zip(plan.fmaps, maps)]
    hm = plan.dense["hm"].to_nchw().cpu()
    print(ds, "prec", prec, "P3", engine.P3, "fmap rel errs", ["%.1e" % e for e in errs], "hm abs err %.2e" % pc.maxabs(hm, out["hm"]), "hm range", float(out["hm"].min()), float(out["hm"].max()),
          "feat absmax %.1f" % float(maps[-1].abs().max()))
