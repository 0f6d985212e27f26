import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
from deft_amd import engine, hiplib, synth
lib = hiplib.get_lib()
sd = synth.synth_state_dict("mot")
g = torch.Generator().manual_seed(3)
for shapes, Q in (((12, 12), 12), ((100,) * 5, 100), ((7, 3, 9), 11)):
    afe1, afe2 = engine.AfePlan(sd, 100, "cuda", lib), engine.AfePlan(sd, 100, "cuda", lib)
    hist = [(torch.rand(n, afe1.D, generator=g) * 3).cuda() for n in shapes]
    cur = (torch.rand(Q, afe1.D, generator=g) * 3).cuda()
    outs = []
    for afe in (afe1, afe1, afe2, afe2):
        outs.append(afe.affinity(hist, cur)[0].clone())
    torch.cuda.synchronize()
    U1 = afe1._work("U", sum(shapes) * 512).clone(); V1 = afe1._work("V", Q * 512).clone()
    U2 = afe2._work("U", sum(shapes) * 512).clone(); V2 = afe2._work("V", Q * 512).clone()
    print(shapes, Q, "fused" if afe1._pair_mlp is not None else "chain", "same plan twice: %.2e | two plans: %.2e | U' %.2e V' %.2e" % (
        float((outs[0] - outs[1]).abs().max()), float((outs[0] - outs[2]).abs().max()), float((U1 - U2).abs().max()), float((V1 - V2).abs().max())))
