"""Why is bench.py's config-B end_to_end slower inside the full default run than alone?  PRE=none|workload|timed|gate python tools/probe/e2e_bisect.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import bench
from deft_amd import hiplib
pre = os.environ.get("PRE", "none")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = hiplib.get_lib()
cfg = bench.CONFIGS["B"]
if pre != "none":
    wl = bench.build_workload(cfg, 32, 2, dev, lib, 0)
    if pre in ("timed", "gate"):
        bench.timed(wl["step"], wl["images"], 5, 2, dev)
    if pre == "gate":
        print(bench.parity_gate(cfg, wl, (0,))["pass_up_to_roundoff_ties"])
    if os.environ.get("DROP", "0") == "1":
        del wl
        torch.cuda.empty_cache()
import gc
if os.environ.get("GC") == "freeze":
    gc.collect(); gc.freeze()
elif os.environ.get("GC") == "off":
    gc.disable()
print("gc objects", len(gc.get_objects()), gc.get_count())
if os.environ.get("PROFILE") == "1":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
e = bench.end_to_end("B", dev, lib, 0, ne=int(os.environ.get("NE", "100")))
if os.environ.get("PROFILE") == "1":
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats("array_tracker|association|kalman|tracker.py|integrate|hiplib|recorder", 45); pstats.Stats(pr).sort_stats("tottime").print_stats(30)
print(pre, os.environ.get("DROP", "0"), e["ms_per_frame"], e["stage_ms"], "| 1-frame", e["one_frame_lookahead"]["ms_per_frame"], "| serial", e["serial"]["ms_per_frame"])
