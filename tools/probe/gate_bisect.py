"""Bisect a failing bench parity gate: config, HIP streams, detections per frame, with / without the profiled step in front."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import deft_oracle as O  # noqa: E402
from deft_amd import hiplib  # noqa: E402


def fmap_errs(wl, cfg, f=0):
    comp, images, sd = wl["comp"], wl["images"], wl["sd"]
    p, j = comp.plans[f // comp.sub], f % comp.sub
    with torch.no_grad():
        out, maps = O.dlaseg_forward(images[f:f + 1].cpu(), sd, cfg["dataset"])
    e = [float((p.fmaps[i].to_nchw()[j].cpu() - m[0]).abs().max()) for i, m in enumerate(maps)]
    return " ".join("%.1e" % v for v in e) + "  hm %.2e" % float((p.dense["hm"].to_nchw()[j].cpu() - out["hm"][0]).abs().max())


def main():
    lib = hiplib.get_lib()
    dev = torch.device("cuda", 0)
    name = sys.argv[1] if len(sys.argv) > 1 else "A"
    for streams, ndet, roof in ((2, None, True), (2, None, False), (1, None, False), (2, 100, False)):
        cfg = dict(bench.CONFIGS[name])
        if ndet is not None:
            cfg["ndet"] = ndet
        wl = bench.build_workload(cfg, 32, streams, dev, lib, 0)
        bench.timed(wl["step"], wl["images"], 6, 2, dev)
        print("config %s streams %d ndet %d after timed loop:        %s" % (name, streams, cfg["ndet"], fmap_errs(wl, cfg)), flush=True)
        if roof:
            bench.roofline_of(wl, lib, 0, 1.0, name)
            print("   after the profiled (serialised) step:           %s" % fmap_errs(wl, cfg), flush=True)
        wl["step"](wl["images"]); torch.cuda.synchronize()
        print("   after one more step:                            %s" % fmap_errs(wl, cfg), flush=True)
        rep = bench.parity_gate(cfg, wl, (0, 17))
        print("   gate:", {k: rep[k] for k in ("pass", "pass_up_to_roundoff_ties", "max_err")}, flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
