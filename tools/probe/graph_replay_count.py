"""Probe (GPU): does the k-th two-branch hipGraph of a process fail to replay?  Detectors built one after the other, each runs a few uint8 frames
(frame 0 eager, frame 1 captures the plan's launch list as a two-branch graph, then replays).  python tools/probe/graph_replay_count.py [n] [keep]
keep=1: the detectors stay alive (their graphs are not destroyed)."""
import gc
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from deft_amd import synth  # noqa: E402
from deft_amd.detector import Detector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
keep = len(sys.argv) > 2 and sys.argv[2] == "1"
sizes = [(96, 128), (128, 160), (64, 96)]
sd = synth.synth_state_dict("mot")
g = np.random.RandomState(0)
alive = []


class Trk:
    def update(self, results, fmaps):
        return []


for i in range(n):
    H, W = sizes[i % len(sizes)]
    opt = SimpleNamespace(dataset="mot", K=20, max_object=100, gpus=[0], hip_graphs=True, depth_scale=1.0, input_h=H, input_w=W, out_thresh=-1.0,
                          test_scales=[1.0], flip_test=False, public_det=False)
    det = Detector(opt, sd)
    det.set_tracker(Trk())
    frames = [g.randint(0, 256, (H + 20, W + 30, 3), dtype=np.uint8) for _ in range(4)]
    for f in frames:
        det.run(f)
    torch.cuda.synchronize()
    sys.stderr.write("detector %d: %d graphs captured, replayed\n" % (i, sum(v is not None for v in det._graphs.values())))
    if keep:
        alive.append(det)
    else:
        del det
        gc.collect()
print("ok", n)
