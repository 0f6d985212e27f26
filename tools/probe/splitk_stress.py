"""Stress of the cross-workgroup split-K hand-over under UNEVEN load: config A's 16-frame sub-batch plans (their 16x16 maps run the
split-K kernels) on two HIP streams, each step started from an idle GPU, so that one plan's split-K launches meet the other plan's
HBM-streaming full-resolution layers.  Every attempt compares frames of both plans with the one-frame plan's result.

    python tools/probe/splitk_stress.py [attempts]      (DEFT_HIP_LIB selects the build)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
from deft_amd import engine, hiplib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = hiplib.get_lib()
    dev = torch.device("cuda", 0)
    bad_total = {}
    t_total = {}
    for name in ("A", "E"):
        cfg = dict(bench.CONFIGS[name])
        wl = bench.build_workload(cfg, 32, 2, dev, lib, 0)
        comp, images = wl["comp"], wl["images"]
        nsplit = sum(1 for _, _, d in comp.plans[0]._gemms if d.splitk > 1)
        refs = {}
        for f in (0, 7, 16, 29):
            p1 = engine.DlaSegPlan(wl["sd"], 1, cfg["H"], cfg["W"], cfg["dataset"], K=100, device="cuda", lib=lib)
            p1.forward(images[f:f + 1]); torch.cuda.synchronize()
            refs[f] = [p1.fmaps[k].to_nchw()[0].clone() for k in (6, 7, 8, 12)]
            del p1
        dt, _ = bench.timed(wl["step"], images, 10, 2, dev)
        bad = 0
        for attempt in range(n):
            torch.cuda.synchronize(); time.sleep(0.01 * (attempt % 4))
            wl["step"](images); torch.cuda.synchronize()
            worst = 0.0
            for f, r in refs.items():
                p, j = comp.plans[f // comp.sub], f % comp.sub
                for k, t in zip((6, 7, 8, 12), r):
                    worst = max(worst, float((p.fmaps[k].to_nchw()[j] - t).abs().max()))
            bad += worst > 0.01
        bad_total[name] = (bad, n, nsplit)
        t_total[name] = dt / 10 * 1e3
        del wl
        torch.cuda.empty_cache()
    # what the protocol costs where split-K matters: one frame per step at config B (hipGraph replay)
    from deft_amd import synth
    p1 = engine.DlaSegPlan(synth.synth_state_dict("mot"), 1, 608, 1088, "mot", K=100, device="cuda", lib=lib)
    x = torch.randn(1, 3, 608, 1088, device="cuda")
    p1.forward(x); torch.cuda.synchronize()
    g = p1.capture_graph()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    t_total["B one frame (graph)"] = (time.perf_counter() - t0) * 10
    print("lib %s: %s   ms/step %s" % (os.path.basename(lib.path), {k: "%d bad of %d (split-K launches per plan: %d)" % v for k, v in bad_total.items()},
                                       {k: round(v, 3) for k, v in t_total.items()}), flush=True)


if __name__ == "__main__":
    main()
