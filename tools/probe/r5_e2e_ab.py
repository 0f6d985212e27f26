"""bench.end_to_end(config) as bench.py runs it (100 timed frames, 50 stored frames in steady state), printed: the A/B harness of the ArrayTracker.begin
switch (DEFT_BEGIN_AHEAD=0 / 1).  python tools/probe/r5_e2e_ab.py [config]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deft_amd import hiplib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "B"
torch.cuda.set_device(0)
out = bench.end_to_end(name, torch.device("cuda", 0), hiplib.get_lib(), 0)
print(json.dumps({"config": name, "begin_ahead": os.environ.get("DEFT_BEGIN_AHEAD", "1"), "ms_per_frame": out["ms_per_frame"], "runs": out["runs_ms_per_frame"],
                  "stage_ms": out["stage_ms"], "one_frame": out["one_frame_lookahead"]["ms_per_frame"], "serial": out["serial"]["ms_per_frame"],
                  "serial_track": out["serial"]["stage_ms"]["track"]}))
