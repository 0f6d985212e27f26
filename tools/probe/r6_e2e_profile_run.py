"""cProfile of the WHOLE per-frame call of the end-to-end loop (Detector._run_once: lookahead bookkeeping, post-processing, the tracker's begin()
and update()) at config B on the GPU box -- r5_e2e_profile.py covers update() alone.  python tools/probe/r6_e2e_profile_run.py [config]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deft_amd import hiplib, detector as DT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "B"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = hiplib.get_lib()
pr = cProfile.Profile()
_run = DT.Detector._run_once
calls = [0]


def prof_run(self, *a, **k):
    calls[0] += 1
    if calls[0] <= 30:
        return _run(self, *a, **k)
    pr.enable()
    try:
        return _run(self, *a, **k)
    finally:
        pr.disable()


DT.Detector._run_once = prof_run
os.environ["DEFT_E2E_REPORTED_MODE_ONLY"] = "1"
out = bench.end_to_end(name, dev, lib, 0, ne=100)
print("profiled run() calls:", calls[0] - 30)
print({k: out[k] for k in ("ms_per_frame", "value", "stage_ms", "tracks_alive")})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
