"""Which call of the tracked stream blocks?  Runs bench.main() with timers around the device-touching calls; prints every call > 3 ms made
from deft_amd/array_tracker.py or deft_amd/detector.py.  DEFT_BENCH_SKIP=... python tools/probe/slow_call_trace.py --no-cpu-baseline --no-check"""
import os, sys, time, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from deft_amd import hiplib
import bench

def wrap(owner, name):
    f = getattr(owner, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        dt = time.perf_counter() - t0
        if dt > 3e-3:
            st = traceback.extract_stack(limit=6)[:-1]
            if any("array_tracker" in s.filename or "detector" in s.filename for s in st):
                sys.stderr.write("SLOW %.2f ms %s.%s <- %s\n" % (dt * 1e3, getattr(owner, "__name__", owner), name,
                                 " <- ".join("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in reversed(st))))
        return r
    setattr(owner, name, g)

for owner, names in [(torch.cuda.Stream, ["synchronize", "wait_event", "wait_stream"]), (torch.cuda.Event, ["synchronize", "record"]),
                     (torch.Tensor, ["to", "copy_", "cpu"]), (torch.cuda.CUDAGraph, ["replay"]), (torch, ["empty", "cat", "zeros"])]:
    for n in names:
        wrap(owner, n)
lib = hiplib.get_lib()
wrap(type(lib), "call")
bench.main()
