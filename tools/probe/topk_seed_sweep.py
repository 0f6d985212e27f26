"""How robust is the bit-exact top-100 index parity at 1088x608?  Sweeps input seeds: HIP path vs the oracle.
Run on the GPU box: python tools/probe/topk_seed_sweep.py [nseeds]"""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import deft_oracle as O
from deft_amd import engine, hiplib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sd = O.synth_state_dict("mot")
H, W = 608, 1088
plan = engine.DlaSegPlan(sd, 1, H, W, "mot", K=100, device="cuda", lib=hiplib.get_lib())
bad = 0
for seed in range(n):
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(100 + seed))
    plan.forward(x.cuda()); torch.cuda.synchronize()
    with torch.no_grad():
        out, _ = O.dlaseg_forward(x, sd, "mot")
    od = O.generic_decode(O.sigmoid_output(out), K=100)
    same = torch.equal(plan.inds[0].cpu().long(), od["inds"][0])
    gaps = (od["scores"][0, :-1] - od["scores"][0, 1:])
    print("seed %d: indices %s  max|dscore| %.2e  max|dbbox| %.2e  min adjacent gap %.2e" % (
        100 + seed, "equal" if same else "DIFFER (%d)" % int((plan.inds[0].cpu().long() != od["inds"][0]).sum()),
        float((plan.scores.cpu() - od["scores"]).abs().max()), float((plan.bboxes.cpu() - od["bboxes"]).abs().max()), float(gaps.min())), flush=True)
    bad += not same
print("%d of %d frames with a different top-100 order" % (bad, n))
