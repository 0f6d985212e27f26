"""cProfile of the end-to-end loop of bench.py (config B: camera frame in host memory -> tracks out) on the GPU box: where the host time of
ArrayTracker.update / FeatureRecorder goes.  python tools/probe/r5_e2e_profile.py [config]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deft_amd import hiplib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "B"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = hiplib.get_lib()
pr = cProfile.Profile()
from deft_amd import array_tracker as AT  # noqa: E402
_upd = AT.ArrayTracker.update
calls = [0]


def prof_update(self, *a, **k):          # profile ONLY the tracker's update() (the host-bound stage of the loop), skipping the warm-up frames
    calls[0] += 1
    if calls[0] <= 60:
        return _upd(self, *a, **k)
    pr.enable()
    try:
        return _upd(self, *a, **k)
    finally:
        pr.disable()


AT.ArrayTracker.update = prof_update
out = bench.end_to_end(name, dev, lib, 0, ne=60)
print("profiled update() calls:", calls[0] - 60)
print({k: out[k] for k in ("ms_per_frame", "value", "stage_ms", "tracks_alive")})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(30)
