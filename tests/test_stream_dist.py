"""-m "not gpu": the sharded tracker stream (deft_amd/stream.py, SURVEY.md §8(e), BASELINE configs[2]).

  * 2 gloo ranks + the REFERENCE's own `Tracker` on rank 0 (needs /root/reference): track ids and boxes must be identical to
    the single-process Level-0 loop `tracker.update(results, FeatureMaps)` over the same frames -- with a cheap deterministic
    stand-in for detection / embedding / affinity, so the test exercises the exchange, the replicated history (incl. an empty
    frame and an object that disappears for two frames) and the replay of the reference tracker, not the kernels;
  * the same on ONE process with the real kernels (DeftModel + AfeSeam through the SIMT emulator) and the accelerated
    tracker forms (one affinity chain per frame, device-side similarity medians)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
HAVE_REF = os.path.isdir("/root/reference/src/lib")
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="/root/reference only exists in the build container")

D, H, W = 12, 60, 80


def _boxes(t):
    """A small scene: 5 objects drifting, one leaves for two frames, frame 4 is empty."""
    if t == 4:
        return []
    out = []
    for i in range(5):
        if i == 3 and t in (6, 7):
            continue
        x0 = 5.0 + 14.0 * i + 0.9 * t * (1 if i % 2 else -0.4)
        y0 = 6.0 + 3.0 * i + 0.6 * t
        w, h = 8.0 + 0.4 * i, 13.0 + 0.5 * i + 0.1 * t
        out.append({"score": 0.9 - 0.04 * i, "class": 1, "bbox": np.array([x0, y0, x0 + w, y0 + h], np.float32)})
    return out


class FakeAFE:
    """Deterministic stand-in for model.AFE: embedding = smooth function of the centre, affinity = softmax-like similarity."""
    host_copy = True
    last_device = None

    def forward_feature_extracter(self, FeatureMaps, centers):
        c = centers.reshape(-1, 2).double()
        k = torch.arange(1, D // 2 + 1, dtype=torch.float64)
        e = torch.cat([torch.sin(c[:, :1] * k * 1.7 + c[:, 1:] * 0.3), torch.cos(c[:, 1:] * k * 1.3 - c[:, :1] * 0.2)], 1)
        return e.float().unsqueeze(0)

    def _block(self, xp, xn):
        d = torch.cdist(xp.double(), xn.double())
        s = torch.exp(-4.0 * d)
        miss = torch.full((xp.shape[0], 1), 0.05, dtype=torch.float64)
        y = torch.cat([s, miss], 1)
        return (y / y.sum(1, keepdim=True)).float()

    def affinity_many(self, hist, cur):
        blocks = [self._block(h, cur) for h in hist]
        starts = [0]
        for b in blocks:
            starts.append(starts[-1] + b.shape[0])
        self.last_device = (torch.cat(blocks, 0), starts)
        return [b.numpy() for b in blocks]

    def forward_stacker_features(self, xp, xn, fill_up_column=True):
        y = self._block(xp[0], xn[0]).numpy()
        P, Q = xp.shape[1], xn.shape[1]
        if fill_up_column and P > 1:
            y = np.concatenate([y, np.repeat(y[:, Q:Q + 1], P - 1, axis=1)], axis=1)
        return y


def _reference_tracker(model):
    import make_golden as MG
    import ref_import
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]            # utils/tracker.py:139 parses argv at import
    try:
        from opts import opts
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
    finally:
        sys.argv = argv
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
    BaseTrack._count = 0
    return RT, RT.Tracker(opt, model, h=H, w=W)


def _log(targets):
    return sorted((int(t.track_id), [round(float(v), 4) for v in t.tlwh]) for t in targets)


def _run_sharded(rank, world, nframes, port, q):
    torch.set_grad_enabled(False)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from deft_amd.stream import ShardedStream
    afe = FakeAFE()
    tracker = None
    if rank == 0:
        _, tracker = _reference_tracker(types.SimpleNamespace(AFE=FakeAFE()))
    st = ShardedStream(lambda f: (_boxes(int(f[0, 0, 0, 0])), None), afe, D, tracker=tracker, dataset="mot", kmax=8, img_h=H, img_w=W,
                       batch=1, device="cpu", snapshot=_log)
    out = []
    for s in range(nframes // world):
        t = s * world + rank
        res = st.step([torch.full((1, 3, 2, 2), float(t))])
        out += res
    q.put((rank, out, st.bytes_gathered))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _collect(world, nframes, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_run_sharded, args=(r, world, nframes, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, out, nbytes = q.get(timeout=300)
        got[r] = (out, nbytes)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_two_ranks_reference_tracker_identical_to_single_process():
    nframes = 10
    torch.set_grad_enabled(False)
    try:
        RT, trk = _reference_tracker(types.SimpleNamespace(AFE=FakeAFE()))
        level0 = [(t, _log(trk.update(_boxes(t), None if not _boxes(t) else [torch.zeros(1, 1, 1, 1)]))) for t in range(nframes)]
    finally:
        torch.set_grad_enabled(True)
    assert sum(len(f) for _, f in level0) > 10 and len({i for _, f in level0 for i, _ in f}) >= 5, "the scene must produce tracks"
    two = _collect(2, nframes, 29731)
    assert two[1][0] == [] and two[0][1] > 0                      # only rank 0 associates; the collectives ran
    assert two[0][0] == level0
    one = _collect(1, nframes, 0)                                 # the same code on one rank, no process group
    assert one[0][0] == level0


@pytest.mark.slow
def test_stream_with_real_kernels_matches_level0_loop(emu_lib):
    """World size 1, real seams: DeftModel (seam 6) + AfeSeam (seams 2, 3) on the emulator build, reference Tracker with
    deft_amd.tracker.accelerate -- ShardedStream must reproduce the plain Level-0 loop exactly."""
    import deft_oracle as O
    from deft_amd import integrate, tracker as DT
    from deft_amd.stream import ShardedStream
    torch.set_grad_enabled(False)
    undo = None
    try:
        sd = O.synth_state_dict("mot")
        model = integrate.DeftModel(sd, "mot", K=20, max_object=100, device="cpu", lib=emu_lib)
        RT, _ = _reference_tracker(model)
        from model.decode import generic_decode
        from opts import opts
        from utils.basetrack import BaseTrack
        argv, sys.argv = sys.argv, ["test.py", "tracking"]
        try:
            opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
        finally:
            sys.argv = argv
        undo = DT.accelerate(RT)
        Hh, Ww, T = 32, 64, 3
        frames = [torch.randn(1, 3, Hh, Ww, generator=torch.Generator().manual_seed(40 + t)) for t in range(T)]

        def detect(x):
            out, fmaps = model(x, None, None)
            out = dict(out[-1]); out["hm"] = out["hm"].sigmoid()
            dets = {k: v.detach().cpu() for k, v in generic_decode(out, K=20, opt=opt).items()}
            thr = float(np.sort(dets["scores"][0].numpy())[::-1][5])
            res = []
            for i in range(dets["scores"].shape[1]):
                if float(dets["scores"][0, i]) < thr:
                    break
                res.append({"score": float(dets["scores"][0, i]), "class": int(dets["clses"][0, i]) + 1,
                            "bbox": dets["bboxes"][0, i].numpy().astype(np.float32) * 4.0})
            return res, fmaps

        BaseTrack._count = 0
        trk = RT.Tracker(opt, model, h=Hh, w=Ww)
        level0 = []
        for t, x in enumerate(frames):
            res, fmaps = detect(x)
            level0.append((t, _log(trk.update(res, fmaps))))
        BaseTrack._count = 0
        trk2 = RT.Tracker(opt, model, h=Hh, w=Ww)
        afe0 = model.AFE                                   # (ShardedStream swaps the tracker's model.AFE -- here the shared model's -- for its replay)
        st = ShardedStream(detect, model.AFE, model.AFE.plan.D, tracker=trk2, dataset="mot", kmax=20, img_h=Hh, img_w=Ww, batch=1, device="cpu", snapshot=_log)
        got = []
        for x in frames:
            got += st.step([x])
        assert got == level0 and sum(len(f) for _, f in level0) > 0
        # the repository's own 2-D tracker on the association rank (what run_stream.py uses where the reference tree is absent): same tracks
        from deft_amd import array_tracker as MT
        model.AFE = afe0
        MT.TrackIds.count = 0
        trk3 = MT.Tracker2D(types.SimpleNamespace(dataset="mot", track_buffer=30, max_object=100, lstm=False), types.SimpleNamespace(AFE=model.AFE),
                            h=Hh, w=Ww)
        st3 = ShardedStream(detect, model.AFE, model.AFE.plan.D, tracker=trk3, dataset="mot", kmax=20, img_h=Hh, img_w=Ww, batch=1, device="cpu", snapshot=_log)
        assert trk3.lazy_blocks is False
        got3 = []
        for x in frames:
            got3 += st3.step([x])
        assert got3 == level0
    finally:
        if undo:
            undo()
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)
