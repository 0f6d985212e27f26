"""Debug aid (GPU): where does a frame's result start to depend on the batch it runs in?  Compares the stage outputs of a 1-frame
and a 2-frame plan (same frame first) with the cross-workgroup split-K off and every tile gate open, every plan-owned buffer in
allocation order (mapped back to the launch that produced it), each plan against its own re-runs, host-serialised launches, and
which rows of which DCN tiles are off.  This is the tool that pinned the intermittent fault of round 2's weight-DMA form of the
igemm.hip DCN (that form is gone).  PROBE_SET=NAME=value,... flips engine switches (e.g. DCN_PATCH=0)."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import deft_oracle as O  # noqa: E402
from deft_amd import engine, hiplib  # noqa: E402

lib = hiplib.get_lib()
sd = O.synth_state_dict("mot")
H, W = 608, 1088
x = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(0))
engine.SPLITK = False
engine.P3_MIN_TILES = 0
engine.DCN_PATCH_MIN_TILES = 0
for kv in os.environ.get("PROBE_SET", "").split(","):          # e.g. PROBE_SET=P3_HALO16=0,FOLD=0
    if kv:
        k, v = kv.split("="); setattr(engine, k, type(getattr(engine, k))(int(v)))
_alloc = engine._Plan.alloc
def alloc(self, *a, **k):
    v = _alloc(self, *a, **k)
    self.__dict__.setdefault("_born", {})[id(v.buf)] = len(self.ops)        # allocated just before op #len(ops) is added
    return v
engine._Plan.alloc = alloc
q1 = engine.DlaSegPlan(sd, 1, H, W, "mot", K=100, device="cuda", lib=lib)
q2 = engine.DlaSegPlan(sd, 2, H, W, "mot", K=100, device="cuda", lib=lib)
q1.forward(x[:1].cuda()); q2.forward(x.cuda()); torch.cuda.synchronize()


def stages(p):
    out = [("base%d" % i, v) for i, v in enumerate(p.base)] + [("fmap%d" % i, v) for i, v in enumerate(p.fmaps)] + [("feat", p.feat), ("hm", p.dense["hm"])]
    return out


def frame0(v):
    return v.to_nchw()[0].float().cpu()


a1 = [(n, frame0(v)) for n, v in stages(q1)]
for (n, a), (_, v2) in zip(a1, stages(q2)):
    b = frame0(v2)
    d = (a - b).abs()
    print("%-8s %-22s N=1 vs N=2: max diff %.3e  (%d elements differ)" % (n, tuple(a.shape), float(d.max()), int((d > 0).sum())), flush=True)
q1.forward(x[:1].cuda()); torch.cuda.synchronize()
for (n, a), (_, v) in zip(a1, stages(q1)):
    d = (a - frame0(v)).abs()
    if float(d.max()) > 0:
        print("NON-DETERMINISTIC %-8s max diff %.3e (%d elements)" % (n, float(d.max()), int((d > 0).sum())))
print("inds equal:", torch.equal(q1.inds[0].cpu(), q2.inds[0].cpu()))

# every plan-owned fp32 buffer in allocation order: frame 0 is the first half of the 2-frame plan's buffer
sizes = {608 * 1088: "608x1088", 304 * 544: "304x544", 152 * 272: "152x272", 76 * 136: "76x136", 38 * 68: "38x68", 19 * 34: "19x34"}
k1 = [t for t in q1._keep if isinstance(t, torch.Tensor) and t.dtype == torch.float32]
k2 = [t for t in q2._keep if isinstance(t, torch.Tensor) and t.dtype == torch.float32]
print(len(k1), len(k2))
shown = 0
for i, (a, b) in enumerate(zip(k1, k2)):
    if b.numel() != 2 * a.numel():
        continue
    d = (a.cpu() - b.cpu()[:a.numel()]).abs()
    nd = int((d > 0).sum())
    if nd:
        hint = [(s, a.numel() // px) for px, s in sizes.items() if a.numel() % px == 0]
        idx = torch.nonzero(d > 0).view(-1)
        ld = hint[0][1] if hint else 1
        px = (idx // ld)
        born = q1._born.get(id(a), -1)
        opn = q1.ops[born][:2] if 0 <= born < len(q1.ops) else None
        print("buffer %3d numel %9d allocated before op %s: %d elements differ, max %.3e; first flat indices %s" % (
            i, a.numel(), opn, nd, float(d.max()), idx[:6].tolist()), flush=True)
        shown += 1
        if shown >= 6:
            break

# which of the two is off?  the same frames through plans whose DCN runs on igemm.hip (weights split in the loop, no DMA in the DCN)
engine.DCN_PATCH = False
r1 = engine.DlaSegPlan(sd, 1, H, W, "mot", K=100, device="cuda", lib=lib)
r2 = engine.DlaSegPlan(sd, 2, H, W, "mot", K=100, device="cuda", lib=lib)
r1.forward(x[:1].cuda()); r2.forward(x.cuda()); torch.cuda.synchronize()
ref = frame0(dict(stages(r1))["fmap6"]); ref2 = frame0(dict(stages(r2))["fmap6"])
print("in-loop-weights plans N=1 vs N=2 equal:", torch.equal(ref, ref2))
for name, pl in (("q1", q1), ("q2", q2)):
    for rep in range(3):
        pl.forward(x[:pl.N].cuda()); torch.cuda.synchronize()
        d = (frame0(dict(stages(pl))["fmap6"]) - ref).abs()
        px = torch.nonzero(d.amax(0) > 0)
        print("%s run %d vs reference: %d elements differ (max %.3e), %d pixels, first %s" % (name, rep, int((d > 0).sum()), float(d.max()), px.shape[0], px[:4].tolist()), flush=True)

# which rows of which 64-pixel DCN tiles are wrong (fmap6 = a 64 -> 64 DCN output at 152x272)
for rep in range(4):
    q2.forward(x.cuda()); torch.cuda.synchronize()
    d = (frame0(dict(stages(q2))["fmap6"]) - ref).abs().amax(0).view(-1)          # per pixel
    bad = torch.nonzero(d > 0).view(-1)
    if bad.numel():
        tiles = {}
        for m in bad.tolist():
            tiles.setdefault(m // 64, []).append(m % 64)
        for t, rows in list(tiles.items())[:8]:
            print("run %d tile %5d: %2d rows wrong: %s" % (rep, t, len(rows), rows), flush=True)
        break

# the same launches with the host waiting after every one: does the difference survive?
def run_serialised(pl, xin):
    pl.image.copy_(xin); torch.cuda.synchronize()
    pl._stream_cache = hiplib.stream_ptr(pl.device)
    for _, _, fn, _ in pl.ops:
        fn(); torch.cuda.synchronize()
    pl._stream_cache = None
for rep in range(5):
    run_serialised(q2, x.cuda())
    d = (frame0(dict(stages(q2))["fmap6"]) - ref).abs()
    print("serialised q2 run %d: %d elements differ" % (rep, int((d > 0).sum())), flush=True)
for rep in range(5):
    q2.forward(x.cuda()); torch.cuda.synchronize()
    d = (frame0(dict(stages(q2))["fmap6"]) - ref).abs().amax(0)
    ys, xs = torch.nonzero(d > 0, as_tuple=True)
    print("async q2 run %d: %d pixels differ%s" % (rep, ys.numel(), "" if not ys.numel() else "; rows %d..%d cols %d..%d" % (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))), flush=True)
