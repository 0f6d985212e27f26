"""Which launch of a BATCHED plan (the sub-batch plans bench.py times: N = 16, pre-split / halo / patch kernels) differs from the
one-frame plan of the same frame (in-loop kernels + split-K: the plan the full-size oracle tests run)?  Per config: the 13 FeatureMaps and
the heat map of frame `f` against the oracle, then every op's written View of that frame, batched plan vs one-frame plan.

    python tools/probe/batch_plan_parity.py [A D E B] [--n 16] [--frame 0]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import deft_oracle as O  # noqa: E402
from deft_amd import engine, hiplib, synth  # noqa: E402
from bench import CONFIGS  # noqa: E402


def record_writes():
    orig = engine._Plan.add

    def add(self, kind, name, fn, flops=0.0, reads=None, writes=None):
        if not hasattr(self, "_wv"):
            self._wv = []
        self._wv.append((kind, name, [v for v in (writes or []) if isinstance(v, engine.View)]))
        return orig(self, kind, name, fn, flops, reads, writes)
    engine._Plan.add = add


def main():
    names = [a for a in sys.argv[1:] if a in CONFIGS] or ["A", "D", "E"]
    N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 16
    f = int(sys.argv[sys.argv.index("--frame") + 1]) if "--frame" in sys.argv else 0
    lib = hiplib.get_lib()
    record_writes()
    for name in names:
        cfg = CONFIGS[name]
        H, W, ds = cfg["H"], cfg["W"], cfg["dataset"]
        sd = synth.synth_state_dict(ds)
        x = torch.randn(32, 3, H, W, generator=torch.Generator().manual_seed(1000))[:N]
        pN = engine.DlaSegPlan(sd, N, H, W, ds, K=100, device="cuda", lib=lib)
        p1 = engine.DlaSegPlan(sd, 1, H, W, ds, K=100, device="cuda", lib=lib)
        pN.forward(x.cuda()); p1.forward(x[f:f + 1].cuda())
        torch.cuda.synchronize()
        a = [v.to_nchw()[f].clone() for v in pN.fmaps]
        pN.forward(x.cuda()); torch.cuda.synchronize()
        rep = max(float((v.to_nchw()[f] - b).abs().max()) for v, b in zip(pN.fmaps, a))
        with torch.no_grad():
            out, maps = O.dlaseg_forward(x[f:f + 1], sd, ds)
        print("== config %s  %dx%d  N=%d frame %d   (repeat-run difference of the batched plan: %.3g)" % (name, W, H, N, f, rep))
        for i, m in enumerate(maps):
            eN = float((pN.fmaps[i].to_nchw()[f].cpu() - m[0]).abs().max())
            e1 = float((p1.fmaps[i].to_nchw()[0].cpu() - m[0]).abs().max())
            print("  fmap %2d %-18s |ref| %.3g   batched err %.3g   one-frame err %.3g" % (i, tuple(m.shape[1:]), float(m.abs().max()), eN, e1))
        eN = float((pN.dense["hm"].to_nchw()[f].cpu() - out["hm"][0]).abs().max())
        e1 = float((p1.dense["hm"].to_nchw()[0].cpu() - out["hm"][0]).abs().max())
        print("  hm logits: batched err %.3g   one-frame err %.3g" % (eN, e1))
        assert len(pN._wv) == len(p1._wv), (len(pN._wv), len(p1._wv))
        shown = 0
        for i, ((kN, nN, vN), (k1, n1, v1)) in enumerate(zip(pN._wv, p1._wv)):
            assert nN == n1, (nN, n1)
            for a_, b_ in zip(vN, v1):
                if (a_.H, a_.W, a_.C) != (b_.H, b_.W, b_.C):
                    continue
                ta, tb = a_.to_nchw()[f], b_.to_nchw()[0]
                err, sc = float((ta - tb).abs().max()), float(tb.abs().max())
                d = pN._op_desc.get(i)
                if err > 2e-4 * max(1.0, sc) and shown < 12:
                    shown += 1
                    extra = "" if d is None else " p3_kernel=%d tile=%#x splitk=%d Cin=%d Cout=%d" % (d.p3_kernel, d.tile, d.splitk, d.Cin, d.Cout)
                    print("  op %3d %-22s %-28s %dx%dx%d  err %.3g (|ref| %.3g)%s" % (i, kN, nN, a_.H, a_.W, a_.C, err, sc, extra))
        del pN, p1
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
