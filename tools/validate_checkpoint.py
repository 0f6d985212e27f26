#!/usr/bin/env python
"""One command for the day a real DEFT checkpoint is at hand (none ships with the reference tree or this image -- README.md:79 points to
external downloads):

    python tools/validate_checkpoint.py <model.pth> [--dataset mot|kitti_tracking|nuscenes] [--lstm <traj.pth>] [--hw 608 1088] [--oracle]

  1. loads it through deft_amd.checkpoint.load_model_state (the reference's load_model rules, model.py:40-110) and prints the key
     coverage: parameters taken, skipped for shape, dropped (not in the dla_34 detector), missing (left at their initial value);
  2. on an MI355X: builds the plan, runs one seeded frame through detection + embedding + affinity, prints score / box statistics and
     checks the outputs are finite;
  3. --oracle: the same frame through oracle/deft_oracle.py on the CPU (test infrastructure, used here as the checker) and the parity
     numbers of the bench's gate (ordered top-K, scores, boxes, embeddings);
  4. --lstm: the trajectory checkpoint through KalmanFilterLSTM's loading rules, one motion step.
Exit status 0 = everything loaded and every check passed."""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def coverage(path, opt):
    from deft_amd import checkpoint as CK
    log = {"Drop": [], "Skip": [], "No param": [], "Reusing": [], "other": []}

    def rec(msg):
        for k in log:
            if msg.startswith(k):
                log[k].append(msg)
                return
        log["other"].append(msg)
    sd = CK.load_model_state(path, opt, log=rec)
    tpl = CK.model_template(opt)
    taken = len(tpl) - len(log["No param"]) - len(log["Skip"])
    print("checkpoint %s" % path)
    for m in log["other"]:
        print("  " + m)
    print("  parameters of the dla_34 detector: %d   taken from the checkpoint: %d   shape-skipped: %d   re-used row-wise: %d   missing: %d   "
          "checkpoint keys dropped: %d" % (len(tpl), taken, len(log["Skip"]), len(log["Reusing"]), len(log["No param"]), len(log["Drop"])))
    for k in ("Skip", "No param", "Drop"):
        for m in log[k][:8]:
            print("    " + m)
        if len(log[k]) > 8:
            print("    ... %d more" % (len(log[k]) - 8))
    return sd, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--dataset", default="mot", choices=["mot", "kitti_tracking", "nuscenes"])
    ap.add_argument("--hw", type=int, nargs=2, default=None, help="network input height width (default: the dataset's bench size)")
    ap.add_argument("--lstm", default="", help="trajectory checkpoint (opt.load_model_traj)")
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle on the frame and print the parity numbers")
    ap.add_argument("--reset-hm", action="store_true"); ap.add_argument("--reuse-hm", action="store_true")
    a = ap.parse_args()
    H, W = a.hw if a.hw else {"mot": (608, 1088), "kitti_tracking": (384, 1280), "nuscenes": (448, 800)}[a.dataset]
    opt = SimpleNamespace(dataset=a.dataset, arch="dla_34", head_conv=256, reset_hm=a.reset_hm, reuse_hm=a.reuse_hm, K=100, max_object=100,
                          gpus=[0], load_model_traj=a.lstm, lstm=bool(a.lstm))
    sd, log = coverage(a.checkpoint, opt)
    ok = not log["Skip"] and not log["No param"]
    if not torch.cuda.is_available():
        print("no GPU here: key coverage only (the plan needs an MI355X)")
        return 0 if ok else 1
    from deft_amd import engine, hiplib
    lib = hiplib.get_lib()
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0))
    plan = engine.DlaSegPlan(sd, 1, H, W, a.dataset, K=100, device="cuda", lib=lib)
    plan.forward(x.cuda())
    afe = engine.AfePlan(sd, 100, "cuda", lib)
    emb = afe.extract(plan.fmaps, plan.centers)
    aff, _ = afe.affinity([emb[0]], emb[0])
    torch.cuda.synchronize()
    s, b = plan.scores[0].cpu(), plan.bboxes[0].cpu()
    fin = bool(torch.isfinite(s).all() and torch.isfinite(b).all() and torch.isfinite(emb).all() and torch.isfinite(aff).all())
    print("one %dx%d frame: top score %.4f, 100th %.4f, box extent %.1f..%.1f px (map), |embedding| max %.3f, self-affinity diagonal mean %.3f, finite: %s"
          % (W, H, float(s[0]), float(s[-1]), float((b[:, 2:] - b[:, :2]).min()), float((b[:, 2:] - b[:, :2]).max()), float(emb.abs().max()),
             float(aff[:, :-1].diagonal().mean()), fin))
    ok = ok and fin
    # the two-fp16-piece arithmetic carries activations of |x| < 4094 (csrc/common.h); beyond that an operand becomes +-inf and the frame NaN:
    # report how far this checkpoint's activations are from that edge (the 13 feature maps cover every stage of the trunk and the neck)
    amax = max(float(fm.buf.abs().max()) for fm in plan.fmaps)
    print("largest |activation| over the 13 feature maps: %.3g (split arithmetic of this library: %d pieces%s)"
          % (amax, lib.pieces, "; fp16 range 4094" if lib.pieces == 2 else ""))
    if lib.pieces == 2 and (not fin or amax >= 1024.0):
        print("WARNING: activations within two binades of the fp16-piece range (or non-finite results): run this checkpoint on the three-bf16-piece "
              "build (DEFT_HIP_LIB=.../deft_amd/lib/libdeft_bf16x3.so), which has no range limit")
    if a.oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import deft_oracle as O
        with torch.no_grad():
            out, maps = O.dlaseg_forward(x, sd, a.dataset)
            od = O.generic_decode(O.sigmoid_output(out), K=100)
        same = torch.equal(plan.inds[0].cpu().long(), od["inds"][0]) and torch.equal(plan.clses[0].cpu().long(), od["clses"][0].long())
        e_s = float((plan.scores[0].cpu() - od["scores"][0]).abs().max()); e_b = float((plan.bboxes[0].cpu() - od["bboxes"][0]).abs().max())
        print("oracle: ordered top-100 identical: %s, score err %.2e, box err %.2e (meaningful when the indices agree)" % (same, e_s, e_b))
        ok = ok and same and e_b <= 1e-3
    if a.lstm:
        from deft_amd import integrate
        kf = integrate.KalmanFilterLSTM(opt, device="cuda", lib=lib)
        h, c = torch.zeros(1, 1, 128), torch.zeros(1, 1, 128)
        _, _, pred = kf.predict(h, c, torch.randn(1, 1, kf.plan.nin))
        print("trajectory checkpoint: %d inputs, %d future steps, finite: %s" % (kf.plan.nin, len(pred), all(np.isfinite(v).all() for v in pred.values())))
    print("OK" if ok else "CHECK THE LINES ABOVE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
