"""End to end with Tracker2D: serial vs one frame of lookahead (GPU).  DEFT_TRACKER_PRIORITY=0/1 python tools/probe/lookahead_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from deft_amd import detector as FD, synth, integrate, array_tracker as MT, hiplib
sd = dict(synth.synth_state_dict("mot"))
sd["ltrb_amodal.2.weight"] = sd["ltrb_amodal.2.weight"] * 0.05
sd["ltrb_amodal.2.bias"] = torch.tensor([-5.0, -8.0, 5.0, 8.0])
H, W = 608, 1088
dev = torch.device("cuda")
opt = SimpleNamespace(dataset="mot", K=100, max_object=100, gpus=[0], hip_graphs=True, depth_scale=1.0, input_h=H, input_w=W,
                      out_thresh=-1.0, test_scales=[1.0], flip_test=False, public_det=False, track_buffer=30, lstm=False)
_junk = []
for _ in range(int(os.environ.get("JUNK_STREAMS", "0"))):       # streams other parts of a process made (and used) before the detector exists
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        _junk.append(torch.zeros(1024, device=dev) + 1)
    _junk.append(st)
torch.cuda.synchronize()
det = FD.Detector(opt, sd)
seam = integrate.AfeSeam(sd, 100, dev, hiplib.get_lib()); seam.host_copy = False
g = np.random.RandomState(0)
NF = 12
_pinned = [torch.from_numpy(g.randint(0, 256, (1080, 1920, 3), dtype=np.uint8)).pin_memory() for _ in range(NF)]
frames = [t.numpy() for t in _pinned] if os.environ.get("PINNED", "1") == "1" else [t.numpy().copy() for t in _pinned]
def e2e(look, n=60):
    MT.TrackIds.count = 0
    det.set_tracker(MT.Tracker2D(opt, SimpleNamespace(AFE=seam), h=1080, w=1920)); det.img_height, det.img_width = 1080, 1920
    npass = look if isinstance(look, int) and look > 1 else 1
    det.lookahead_frames = npass
    def nxt(i, end):
        f = [frames[(i + j) % NF] for j in range(1, 2 * npass) if i + j < end]
        if not look or not f:
            return None
        return f if npass > 1 else f[0]
    for i in range(56):
        det.run(frames[i % NF], prefetch=nxt(i, 56 + n))
    torch.cuda.synchronize(); acc = {}
    t0 = time.perf_counter()
    for i in range(n):
        det.run(frames[(56 + i) % NF], prefetch=nxt(56 + i, 56 + n))
        for k, v in det.times.items(): acc[k] = acc.get(k, 0) + v
    torch.cuda.synchronize()
    print({False: "serial   ", True: "lookahead"}.get(look, "%d / pass" % npass), "%.2f ms/frame" % ((time.perf_counter() - t0) / n * 1e3), {k: round(v / n * 1e3, 2) for k, v in acc.items()}, "tracks", len(det.tracker.tracked_stracks))
for m in [False, True, 4, True, 4]:
    e2e(m)
