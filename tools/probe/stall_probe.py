"""What makes a small launch on another stream wait for a whole lookahead pass?  (GPU)
A 4-frame pass replayed as a hipGraph on a side stream; beside it, on stream X (null / normal / high priority): [optionally wait for an
event recorded on the null stream] -> tiny H2D from pinned -> tiny kernel -> tiny D2H -> synchronize.  Prints the round-trip time of each
variant while the pass is running (idle device: ~0.05 ms)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from deft_amd import detector as FD, synth
dev = torch.device("cuda")
junk = []
for _ in range(int(os.environ.get("JUNK_STREAMS", "0"))):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        junk.append(torch.zeros(1024, device=dev) + 1)
    junk.append(st)
sd = synth.synth_state_dict("mot")
opt = SimpleNamespace(dataset="mot", K=100, max_object=100, gpus=[0], hip_graphs=True, depth_scale=1.0, input_h=608, input_w=1088,
                      out_thresh=-1.0, test_scales=[1.0], flip_test=False, public_det=False, track_buffer=30, lstm=False)
det = FD.Detector(opt, sd)
sl = FD.Detector._Slot(det, 608, 1088, 1080, 1920, n=4)
g = np.random.RandomState(0)
frames = [torch.from_numpy(g.randint(0, 256, (1080, 1920, 3), dtype=np.uint8)).pin_memory().numpy() for _ in range(4)]
for _ in range(3):
    det._launch_ahead(sl, frames); sl.done.synchronize()
assert sl.graph is not None
t0 = time.perf_counter(); det._launch_ahead(sl, frames); sl.done.synchronize()
print("pass alone: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pin_in = torch.zeros(4096, dtype=torch.float32).pin_memory(); pin_out = torch.zeros(4096, dtype=torch.float32).pin_memory()
pageable = torch.zeros(4096)
null = torch.cuda.default_stream(dev)
streams = {"null": null, "normal": torch.cuda.Stream(), "high": torch.cuda.Stream(priority=-1)}

def small(stream, wait_null, src):
    with torch.cuda.stream(stream):
        if wait_null:
            stream.wait_stream(null)
        x = src.to(dev, non_blocking=True)
        y = x * 2 + 1
        pin_out.copy_(y, non_blocking=True)
        stream.synchronize()

for name, st in streams.items():
    for wait_null in (False, True):
        for src_name, src in (("pinned", pin_in), ("pageable", pageable)):
            small(st, wait_null, src)                     # warm
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                det._launch_ahead(sl, frames)
                time.sleep(0.001)                         # the pass is under way
                t0 = time.perf_counter()
                small(st, wait_null, src)
                ts.append((time.perf_counter() - t0) * 1e3)
                sl.done.synchronize()
            print("stream %-6s wait_null=%d src=%-8s: %s ms" % (name, wait_null, src_name, " ".join("%.2f" % t for t in ts)))
