"""Find the first launch whose output is wrong when a step of the 2-stream sub-batch plans starts from an idle GPU (config A)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools", "probe"))
import bench  # noqa: E402
from batch_plan_parity import record_writes  # noqa: E402
from deft_amd import engine, hiplib  # noqa: E402


def main():
    lib = hiplib.get_lib()
    dev = torch.device("cuda", 0)
    name = sys.argv[1] if len(sys.argv) > 1 else "A"
    record_writes()
    cfg = dict(bench.CONFIGS[name])
    for streams in (2, 1):
        wl = bench.build_workload(cfg, 32, streams, dev, lib, 0)
        comp, images = wl["comp"], wl["images"]
        refs = {}
        for f in (0, 16):
            p1 = engine.DlaSegPlan(wl["sd"], 1, cfg["H"], cfg["W"], cfg["dataset"], K=100, device="cuda", lib=lib)
            p1.forward(images[f:f + 1]); torch.cuda.synchronize()
            refs[f] = p1
        bench.timed(wl["step"], images, 4, 1, dev)
        nbad = 0
        for attempt in range(12):
            torch.cuda.synchronize(); time.sleep(0.05 * (attempt % 3))
            wl["step"](images); torch.cuda.synchronize()
            bad = None
            for f in (0, 16):
                if f // comp.sub >= len(comp.plans):
                    continue
                p, j = comp.plans[f // comp.sub], f % comp.sub
                e = float((p.fmaps[8].to_nchw()[j] - refs[f].fmaps[8].to_nchw()[0]).abs().max())
                if e > 0.01:
                    bad = (f, e)
                    break
            print("streams %d attempt %d: %s" % (streams, attempt, "ok" if bad is None else "BAD frame %d fmap8 err %.3g" % bad), flush=True)
            if bad is None:
                continue
            nbad += 1
            if nbad > 2:
                continue
            f = bad[0]
            p, j, p1 = comp.plans[f // comp.sub], f % comp.sub, refs[f]
            shown = 0
            for i, ((kN, nN, vN), (k1, n1, v1)) in enumerate(zip(p._wv, p1._wv)):
                for a_, b_ in zip(vN, v1):
                    if (a_.H, a_.W, a_.C) != (b_.H, b_.W, b_.C):
                        continue
                    # every frame of the sub-batch against ITS one-frame result is too slow: frame j only, but count bad frames cheaply
                    ta, tb = a_.to_nchw()[j], b_.to_nchw()[0]
                    err, sc = float((ta - tb).abs().max()), float(tb.abs().max())
                    d = p._op_desc.get(i)
                    if err > 1e-3 * max(1.0, sc) and shown < 6:
                        shown += 1
                        extra = "" if d is None else " p3_kernel=%d tile=%#x splitk=%d Cin=%d Cout=%d M=%d" % (d.p3_kernel, d.tile, d.splitk, d.Cin, d.Cout, d.M)
                        nz = (ta - tb).abs().amax(0)          # [H, W]
                        ys, xs = torch.nonzero(nz > 1e-3 * max(1.0, sc), as_tuple=True)
                        print("    op %3d %-20s %-30s %dx%dx%d err %.3g (|ref| %.3g)%s  bad pixels %d rows %d..%d cols %d..%d" % (
                            i, kN, nN, a_.H, a_.W, a_.C, err, sc, extra, len(ys), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
