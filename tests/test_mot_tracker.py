"""-m "not gpu" (build container: needs /root/reference): deft_amd.array_tracker.Tracker2D -- the 2-D tracking loop as arrays, on the
device forms -- against the REFERENCE's own `Tracker` frame by frame: same track ids, same boxes, same activation flags, over a
scene with births, a two-frame occlusion, an empty frame, a crossing pair and tracks that age out.  The embedding / affinity side is
a cheap deterministic stand-in (so the association logic is what is compared); tests/test_reference_detector.py and the GPU replay
cover the kernels."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
HAVE_REF = os.path.isdir("/root/reference/src/lib")
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="/root/reference only exists in the build container")

D, H, W = 12, 120, 200


class FakeAFE:
    """Deterministic stand-in for model.AFE with BOTH interfaces: the reference's pairwise `forward_stacker_features` and the
    one-chain `affinity_many` + `last_device` the device recorder uses."""
    host_copy = True
    last_device = None
    plan = None

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


def _scene(t, dataset):
    """Detections of frame t: objects drift; #3 is occluded in frames 6-7; #5 is born at frame 9; #1 leaves for good at 14 (its track must
    age out); frame 4 is empty; #0 and #2 cross around frame 12; KITTI gets a class mix (only class 2 is tracked)."""
    if t == 4 or t >= 26:                 # the stream ends with empty frames: every track ages out (max_time_lost = 10 frames)
        return []
    out = []
    for i in range(7):
        if (i == 3 and t in (6, 7)) or (i == 5 and t < 9) or (i == 1 and t >= 14) or (i == 6 and t % 5 == 0):
            continue
        vx = [1.8, -0.7, -1.6, 0.5, 1.1, -0.9, 0.3][i]
        x0 = [20.0, 60.0, 80.0, 110.0, 140.0, 30.0, 165.0][i] + vx * t
        y0 = 10.0 + 12.0 * i + 0.4 * t * (1 if i % 2 else -1)
        w, h = 14.0 + 0.5 * i, 26.0 + 0.7 * i + 0.05 * t
        jit = 0.3 * np.sin(1.3 * t + i)
        cls = 2 if dataset == "mot" or i % 3 != 2 else 1
        out.append({"score": float(0.9 - 0.05 * i + 0.01 * np.cos(t + i)), "class": cls if dataset != "mot" else 1,
                    "bbox": np.array([x0 + jit, y0 - jit, x0 + w + jit, y0 + h], np.float32)})
    if 2 <= t <= 12:                      # an isolated object that disappears for good: its track ages out (max_time_lost = 10 frames)
        out.append({"score": 0.5, "class": 2 if dataset != "mot" else 1, "bbox": np.array([180.0, 92.0 + 0.1 * t, 193.0, 116.0], np.float32)})
    return out


def _reference(dataset, model):
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
    opt = opts().parse(["tracking", "--dataset", dataset, "--gpus", "-1"])
    BaseTrack._count = 0
    return opt, RT.Tracker(opt, model, h=H, w=W)


def _log(targets):
    return [(int(t.track_id), bool(t.is_activated), int(t.tracklet_len), [float(v) for v in t.tlwh], float(t.score)) for t in targets]


@pytest.mark.parametrize("dataset", ["mot", "kitti_tracking"])
def test_tracker2d_matches_reference_tracker(emu_lib, dataset):
    from deft_amd import array_tracker as MT
    nframes = 40
    opt, ref = _reference(dataset, types.SimpleNamespace(AFE=FakeAFE()))
    MT.TrackIds.count = 0
    afe = FakeAFE()
    afe.plan = types.SimpleNamespace(lib=emu_lib, _stream=lambda: None)       # the similarity medians run through the C ABI (deft_track_similarity)
    mine = MT.Tracker2D(opt, types.SimpleNamespace(AFE=afe), h=H, w=W)
    fm = [torch.zeros(1, 1, 1, 1)]
    ids = set()
    for t in range(nframes):
        res = _scene(t, dataset)
        a = _log(ref.update([dict(r) for r in res], fm))
        b = _log(mine.update([dict(r) for r in res], fm))
        assert [x[:3] for x in a] == [x[:3] for x in b], (t, a, b)
        for x, y in zip(a, b):
            assert np.abs(np.array(x[3]) - np.array(y[3])).max() <= 1e-9 and x[4] == y[4], (t, x, y)
        ids |= {x[0] for x in a}
        assert sorted(tk.track_id for tk in ref.tracked_stracks) == sorted(tk.track_id for tk in mine.tracked_stracks), t
    assert len(ids) >= 5
    # tracks aged out (MOT; on KITTI a track unseen for 6 frames leaves the IoU candidates and with them the only place that removes: tracker.py:982-1010)
    assert len(mine.removed_stracks) == len(ref.removed_stracks) and (dataset != "mot" or len(mine.removed_stracks) >= 5)


def test_kalman_batch_forms_match_reference_filter():
    """kf_initiate / kf_multi_predict / kf_multi_update against utils/tracking_utils/kalman_filter.py (its per-track Cholesky update)."""
    _reference("mot", types.SimpleNamespace(AFE=FakeAFE()))
    from utils.tracking_utils.kalman_filter import KalmanFilter
    from deft_amd import array_tracker as MT
    kf = KalmanFilter()
    g = np.random.default_rng(0)
    meas = np.stack([g.uniform(10, 500, 9), g.uniform(10, 300, 9), g.uniform(0.3, 0.6, 9), g.uniform(40, 200, 9)], 1)
    ms, cs = zip(*[kf.initiate(m) for m in meas])
    for m, c, z in zip(ms, cs, meas):
        m2, c2 = MT.kf_initiate(z)
        assert np.array_equal(m, m2) and np.array_equal(c, c2)
    mean, cov = np.stack(ms), np.stack(cs)
    for step in range(5):
        rm, rc = kf.multi_predict(mean.copy(), cov.copy())
        mm, mc = MT.kf_multi_predict(mean, cov)
        assert np.array_equal(rm, mm) and np.allclose(rc, mc, rtol=0, atol=1e-12)
        z = meas + g.normal(0, 2.0, meas.shape) * [1, 1, 0.01, 1]
        ru = [kf.update(rm[i], rc[i], z[i]) for i in range(len(z))]
        um, uc = MT.kf_multi_update(mm, mc, z)
        assert np.allclose(np.stack([u[0] for u in ru]), um, rtol=0, atol=1e-9) and np.allclose(np.stack([u[1] for u in ru]), uc, rtol=0, atol=1e-9)
        mean, cov = um, uc


def test_inline_kalman_gate_equals_fuse_motion():
    """Tracker2D.update gates with association._maha2 on its stacked arrays instead of calling association.fuse_motion track by track:
    the two are the same expressions -- bit for bit, including the gated (inf) entries."""
    from types import SimpleNamespace
    from deft_amd import association as A
    g = np.random.RandomState(2)
    T, N = 37, 23
    mean = np.concatenate([g.rand(T, 2) * 300, g.rand(T, 1) * 0.5 + 0.2, g.rand(T, 1) * 80 + 20, g.randn(T, 4)], 1)
    a = g.randn(T, 8, 8) * 3
    cov = a @ a.transpose(0, 2, 1) + np.eye(8) * 5
    xyah = np.concatenate([g.rand(N, 2) * 300, g.rand(N, 1) * 0.5 + 0.2, g.rand(N, 1) * 80 + 20], 1)
    xyah[:5, :2] = mean[:5, :2] + g.randn(5, 2)                      # some detections inside the gate
    cost = g.rand(T, N)
    tracks = [SimpleNamespace(mean=mean[t], covariance=cov[t]) for t in range(T)]
    dets = [SimpleNamespace(to_xyah=(lambda r=xyah[j]: r.copy())) for j in range(N)]
    ref = A.fuse_motion(None, cost.copy(), tracks, dets, frame_id=3, use_lstm=False)
    d = cost.copy()
    gate = A._maha2(mean[:, :2], cov[:, :2, :2], xyah[:, :2])
    d[gate > 5.0 * A.chi2inv95[2]] = np.inf
    d = 0.9 * d + 0.05 * (1 - 0.9) * gate
    assert np.array_equal(ref, d) and np.isinf(d).any() and np.isfinite(d).any()


def test_lazy_affinity_blocks_change_nothing(emu_lib):
    """Tracker2D.lazy_blocks: the new frame is scored only against the stored frames that hold one of the pool's selected nodes (the
    reference scores all of them, tracker.py:76-90, and reads a few).  A crowded random scene with drop-outs and re-finds: identical
    outputs frame by frame with and without, and the lazy run really asked for fewer blocks."""
    from deft_amd import array_tracker as MT
    opt = types.SimpleNamespace(dataset="mot", track_buffer=30, max_object=100, lstm=False)
    g = np.random.RandomState(5)
    base = np.concatenate([g.rand(30, 2) * np.array([170.0, 90.0]), g.rand(30, 2) * 8 + 10], 1)
    vel = g.randn(30, 2) * 0.6
    gone = {i: (int(g.randint(5, 40)), int(g.randint(2, 9))) for i in range(0, 30, 3)}        # object -> (first missing frame, length)
    frames = []
    for t in range(55):
        rows = []
        for i in range(30):
            if i in gone and gone[i][0] <= t < gone[i][0] + gone[i][1]:
                continue
            x, y = base[i, :2] + vel[i] * t
            w, h = base[i, 2:]
            rows.append({"score": float(0.6 + 0.01 * i), "class": 1, "bbox": np.array([x, y, x + w, y + h], np.float32)})
        frames.append(rows)

    def run(lazy):
        MT.TrackIds.count = 0
        afe = FakeAFE()
        afe.plan = types.SimpleNamespace(lib=emu_lib, _stream=lambda: None)
        asked = []
        many = afe.affinity_many
        afe.affinity_many = lambda hist, cur: (asked.append(len(hist)), many(hist, cur))[1]
        trk = MT.Tracker2D(opt, types.SimpleNamespace(AFE=afe), h=H, w=W)
        trk.lazy_blocks = lazy
        return [_log(trk.update([dict(r) for r in rows], [torch.zeros(1)])) for rows in frames], asked, trk

    full, asked_full, _ = run(False)
    lazy, asked_lazy, trk = run(True)
    assert full == lazy and sum(len(f) for f in full) > 1000
    assert max(asked_full) == 49 and sum(asked_lazy) < 0.4 * sum(asked_full)
    assert len(trk.lost_stracks) + len(trk.removed_stracks) >= 0 and MT.TrackIds.count >= 30


@pytest.mark.parametrize("dataset", ["mot", "kitti_tracking"])
def test_native_association_equals_the_numpy_stages(emu_lib, dataset):
    """ArrayTracker.native_assoc: the cascade of a frame through deft_associate_2d + deft_kf_predict / deft_kf_update (one host call each) against the
    numpy stages (which the reference-tracker tests above pin) on the crowded random scene with drop-outs and re-finds: the same tracks in the same
    order frame by frame (ids, activation, length, score identical; boxes to 1e-9: the native Kalman update sums in another order than BLAS)."""
    from deft_amd import array_tracker as MT
    opt = types.SimpleNamespace(dataset=dataset, track_buffer=30, max_object=100, lstm=False)
    g = np.random.RandomState(11)
    base = np.concatenate([g.rand(30, 2) * np.array([170.0, 90.0]), g.rand(30, 2) * 8 + 10], 1)
    vel = g.randn(30, 2) * 0.6
    gone = {i: (int(g.randint(5, 40)), int(g.randint(2, 9))) for i in range(0, 30, 3)}
    frames = []
    for t in range(55):
        rows = []
        for i in range(30):
            if i in gone and gone[i][0] <= t < gone[i][0] + gone[i][1]:
                continue
            x, y = base[i, :2] + vel[i] * t
            w, h = base[i, 2:]
            rows.append({"score": float(0.6 + 0.01 * i), "class": 2 if i % 4 else 1, "bbox": np.array([x, y, x + w, y + h], np.float32)})
        frames.append(rows)

    def run(native):
        MT.TrackIds.count = 0
        afe = FakeAFE()
        calls = []
        lib = types.SimpleNamespace(call=lambda name, *a: (calls.append(name), emu_lib.call(name, *a))[1])
        afe.plan = types.SimpleNamespace(lib=lib, _stream=lambda: None)
        trk = MT.Tracker2D(opt, types.SimpleNamespace(AFE=afe), h=H, w=W)
        trk.native_assoc = native
        return [_log(trk.update([dict(r) for r in rows], [torch.zeros(1)])) for rows in frames], calls

    a, calls_a = run(True)
    b, calls_b = run(False)
    assert "deft_associate_2d" in calls_a and "deft_kf_update" in calls_a and "deft_associate_2d" not in calls_b
    assert sum(len(f) for f in a) > 500
    for t, (fa, fb) in enumerate(zip(a, b)):
        assert [x[:3] + (x[4],) for x in fa] == [x[:3] + (x[4],) for x in fb], t
        for x, y in zip(fa, fb):
            assert np.abs(np.array(x[3]) - np.array(y[3])).max() <= 1e-9, (t, x, y)


@pytest.mark.parametrize("mm", [4, 2])
def test_native_node_selection_and_gather_table_equal_numpy(emu_lib, mm):
    """deft_track_nodes (assoc.hip) against ArrayTracker's numpy restatement of STrack.get_similarity's node selection (tracker.py:221-248) and of
    the gather table deft_track_similarity reads: random node tables with young / old / short histories -- the same mask, the same rows | scale |
    cnt record bit for bit; a selected node whose frame has no block is the reference's KeyError(frame), an id past its block an IndexError, the
    same offender as numpy's row-major scan reports."""
    from deft_amd import array_tracker as MT
    g = np.random.RandomState(5 + mm)
    L, T, fid, F = mm + 2, 90, 200, 60

    def tracker(native, nf, ni, nn, index, starts, nd):
        t = MT.ArrayTracker.__new__(MT.ArrayTracker)
        t.mm, t.native_assoc = mm, native
        cols = {"nf": nf, "ni": ni, "nn": nn}
        t.cols = types.SimpleNamespace(__getitem__=None)
        t.cols = type("C", (), {"__getitem__": lambda s_, k: cols[k]})()
        seen = []

        def call(name, *a):                                   # deft_track_similarity: keep the gather table it was handed (rows | scale | cnt)
            import ctypes as C
            assert name == "deft_track_similarity"
            seen.append(np.ctypeslib.as_array(C.cast(a[3].value, C.POINTER(C.c_int32)), (2 * T * L + T,)).copy())
            return 0
        lib = types.SimpleNamespace(_fn=emu_lib._fn, last_error=emu_lib.last_error, call=call)
        t.model = types.SimpleNamespace(AFE=types.SimpleNamespace(plan=types.SimpleNamespace(lib=lib, _stream=lambda: None)))
        t.recorder = types.SimpleNamespace(_dev=(fid, torch.zeros(int(starts[-1]), nd + 1), starts, index))
        return t, seen

    def record(t, seen, nd):
        sel_all = t._selected_nodes(fid)
        t._similarity(fid, np.arange(T), nd, sel_all)
        return sel_all[2].copy(), seen[-1]

    for trial in range(6):
        nn = g.randint(0, 12, T).astype(np.int64)
        nf = np.sort(g.randint(fid - F, fid, (T, L)), 1).astype(np.int64)
        if trial % 2:
            nf[:, -3:] = np.sort(g.randint(fid - 6, fid, (T, 3)), 1)                 # mostly young, some nodes just past max_track_node
        blocks = sorted(set(nf.reshape(-1).tolist()))
        lens = g.randint(1, 9, len(blocks))
        starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        index = {f: (k, np.float32(0.5 + 0.01 * k)) for k, f in enumerate(blocks)}
        ni = np.stack([[g.randint(0, lens[index[f][0]]) for f in row] for row in nf.tolist()]).astype(np.int64)
        a, sa = tracker(True, nf, ni, nn, index, starts, 7)
        b, sb = tracker(False, nf, ni, nn, index, starts, 7)
        ma, ra = record(a, sa, 7)
        mb, rb = record(b, sb, 7)
        assert np.array_equal(ma, mb) and ma.dtype == mb.dtype == np.bool_ and np.array_equal(ra, rb), trial
        assert ma.any() and not ma.all()
        # a selected node whose frame has no block / whose id is past its block
        tt, cc = np.argwhere(mb)[len(np.argwhere(mb)) // 2]
        miss = dict(index); gone = int(nf[tt, cc]); del miss[gone]
        for native in (True, False):
            t_, s_ = tracker(native, nf, ni, nn, miss, starts, 7)
            with pytest.raises(KeyError) as e:
                record(t_, s_, 7)
            assert e.value.args[0] == gone
            ni2 = ni.copy(); ni2[tt, cc] = 99
            t_, s_ = tracker(native, nf, ni2, nn, index, starts, 7)
            with pytest.raises(IndexError):
                record(t_, s_, 7)


@pytest.mark.parametrize("dataset", ["mot", "nuscenes_2d_rule"])
def test_prepare_ahead_covers_every_block_the_next_frame_reads(emu_lib, dataset):
    """ArrayTracker.prepare scores frame k + 2 before update(k + 1) has run, against a SUPERSET of the stored frames that update will leave selected
    (today's table evaluated at the later frame number, plus the begun frame).  A crowded scene with drop-outs of 2-8 frames and re-finds, long
    enough for nodes to age out of the 50-frame window (the case where a track's selection GROWS back from "the last mm" to "all mm + 1"): no
    KeyError for a missing block, and the same tracks as plain update() calls."""
    from deft_amd import array_tracker as MT
    opt = types.SimpleNamespace(dataset="mot", track_buffer=30 if dataset == "mot" else 240, max_object=100, lstm=False)
    g = np.random.RandomState(23)
    n = 24
    base = np.concatenate([g.rand(n, 2) * np.array([170.0, 90.0]), g.rand(n, 2) * 8 + 10], 1)
    vel = g.randn(n, 2) * 0.3
    frames = []
    for t in range(150):
        rows = []
        for i in range(n):
            period, off, out = 9 + i, 3 * i, 2 + i % 7            # every object drops out for `out` frames once per `period` (+ one long absence)
            if (t + off) % period < out or (i % 5 == 0 and 60 <= t < 60 + 40 + i):
                continue
            x, y = base[i, :2] + vel[i] * t
            w, h = base[i, 2:]
            rows.append({"score": float(0.6 + 0.01 * i), "class": 1, "bbox": np.array([x, y, x + w, y + h], np.float32)})
        frames.append(rows)

    def run(two):
        MT.TrackIds.count = 0
        afe = FakeAFE()
        afe.plan = types.SimpleNamespace(lib=emu_lib, _stream=lambda: None)
        trk = MT.Tracker2D(opt, types.SimpleNamespace(AFE=afe), h=H, w=W)
        fm = [torch.zeros(1, 1, 1, 1)]
        log = []
        for t in range(len(frames)):
            log.append(_log(trk.update(frames[t], fm)))
            if two and t + 1 < len(frames):
                trk.begin(frames[t + 1], fm)
                if t + 2 < len(frames):
                    trk.prepare(frames[t + 2], fm)
        return log

    plain = run(False)
    assert run(True) == plain
    assert sum(len(f) for f in plain) > 1500 and max(x[2] for f in plain for x in f) > 20


@pytest.mark.parametrize("dataset", ["mot", "kitti_tracking"])
def test_begin_ahead_changes_nothing(emu_lib, dataset):
    """ArrayTracker.begin(results, FeatureMaps): the device half of the NEXT frame queued behind update(k).  Same tracks as plain update() calls when
    every frame is begun ahead; and a begin() for a frame that never comes (update() is handed another frame's detections) is taken back -- the
    recorder's stored frames, evicted entries included (a stream longer than its 50-frame window), are what they were."""
    from deft_amd import array_tracker as MT
    opt = types.SimpleNamespace(dataset=dataset, track_buffer=30, max_object=100, lstm=False)
    nframes = 64
    frames = [[dict(r) for r in _scene(t % 24, dataset)] for t in range(nframes)]
    wrong = [dict(r) for r in _scene(3, dataset)][:2]

    def run(mode):
        MT.TrackIds.count = 0
        afe = FakeAFE()
        afe.plan = types.SimpleNamespace(lib=emu_lib, _stream=lambda: None)
        trk = MT.Tracker2D(opt, types.SimpleNamespace(AFE=afe), h=H, w=W)
        fm = [torch.zeros(1, 1, 1, 1)]
        log = []
        for t in range(nframes):
            log.append(_log(trk.update(frames[t], fm)))
            if mode == "ahead" and t + 1 < nframes:
                trk.begin(frames[t + 1], fm)
            elif mode == "wrong" and t + 1 < nframes:
                trk.begin(wrong if t % 3 else frames[t + 1], fm)       # two of three announcements are for a frame that never comes
            elif mode == "two" and t + 1 < nframes:                    # begin(k + 1) behind update(k), prepare(k + 2) behind that: Detector.run's order
                trk.begin(frames[t + 1], fm)
                if t + 2 < nframes:
                    trk.prepare(frames[t + 2], fm)
                    assert trk._prepared["fid"] == t + 3 and (not len(frames[t + 2]) or t + 3 in trk.recorder.all_features)
            elif mode == "two_wrong" and t + 1 < nframes:              # ... with announcements that do not come true, at either distance
                trk.begin(wrong if t % 4 == 1 else frames[t + 1], fm)
                if t + 2 < nframes:
                    trk.prepare(wrong if t % 4 == 2 else frames[t + 2], fm)
                if t % 4 == 3 and t + 2 < nframes:
                    trk.prepare(frames[t + 2], fm)                     # prepared twice: the first is taken back
            elif mode == "prepare_only" and t + 1 < nframes:           # prepare() with nothing begun = the first part of the next frame's begin()
                trk.prepare(frames[t + 1], fm)
                if t % 2:
                    trk.begin(frames[t + 1], fm)
        assert trk._begun is None and trk._prepared is None
        return log, sorted(trk.recorder.all_features), trk.frame_id

    plain = run("plain")
    assert run("ahead") == plain and run("wrong") == plain
    assert run("two") == plain and run("two_wrong") == plain and run("prepare_only") == plain
    assert sum(len(f) for f in plain[0]) > 200 and len(plain[1]) == 50 and plain[2] == nframes


def test_native_kalman_matches_reference_filter(emu_lib):
    """deft_kf_predict / deft_kf_update against utils/tracking_utils/kalman_filter.py: predict bit for bit, update to round-off; a projected
    covariance that is not positive definite is an error (-94), as numpy's Cholesky raises."""
    import ctypes as C
    _reference("mot", types.SimpleNamespace(AFE=FakeAFE()))
    from utils.tracking_utils.kalman_filter import KalmanFilter
    kf = KalmanFilter()
    g = np.random.default_rng(3)
    meas = np.stack([g.uniform(10, 500, 9), g.uniform(10, 300, 9), g.uniform(0.3, 0.6, 9), g.uniform(40, 200, 9)], 1)
    ms, cs = zip(*[kf.initiate(m) for m in meas])
    mean, cov = np.stack(ms), np.stack(cs)
    p = lambda a: C.c_void_p(a.ctypes.data)
    for step in range(6):
        rm, rc = kf.multi_predict(mean.copy(), cov.copy())
        emu_lib.call("deft_kf_predict", p(mean), p(cov), len(mean))
        assert np.array_equal(rm, mean) and np.array_equal(rc, cov)
        rows = np.ascontiguousarray(g.permutation(9)[:6], dtype=np.int32)
        z = np.ascontiguousarray((meas + g.normal(0, 2.0, meas.shape) * [1, 1, 0.01, 1])[rows])
        for k, i in enumerate(rows.tolist()):
            rm[i], rc[i] = kf.update(rm[i], rc[i], z[k])
        emu_lib.call("deft_kf_update", p(mean), p(cov), p(rows), len(rows), p(z))
        assert np.allclose(rm, mean, rtol=0, atol=1e-9) and np.allclose(rc, cov, rtol=0, atol=1e-9)
        untouched = np.setdiff1d(np.arange(9), rows)
        assert np.array_equal(rm[untouched], mean[untouched])
    bad = cov.copy()
    bad[2, :4, :4] = -np.eye(4) * 1e6
    with pytest.raises(Exception, match="not positive definite"):
        emu_lib.call("deft_kf_update", p(mean), p(bad), p(np.array([2], np.int32)), 1, p(np.ascontiguousarray(meas[:1])))


# ---------------------------------------------------------------------------------------------------------------------------------------
# round 4: the LSTM configuration (BASELINE configs[3]) and the nuScenes 3-D association (configs[4]) of the array tracker, against the
# reference's own Tracker with its own KalmanFilterLSTM (synthetic LSTM weights on both sides)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _reference_lstm(dataset, model, lstm, lsd):
    import make_golden as MG
    import ref_import
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
    finally:
        sys.argv = argv
    opt = opts().parse(["tracking", "--dataset", dataset, "--gpus", "-1"])
    opt.lstm = lstm
    ref_cls = RT.KalmanFilterLSTM

    def make_ref(o):
        k = ref_cls(o)
        k.model.load_state_dict(lsd, strict=True)
        k.model.eval()
        return k
    if lstm:
        RT.KalmanFilterLSTM = make_ref
        RT.STrack.shared_kalman_lstm = make_ref(opt)
    BaseTrack._count = 0
    trk = RT.Tracker(opt, model, h=H, w=W)
    assert trk.use_lstm == lstm
    return opt, trk, (RT, ref_cls)


def _mine(opt, emu_lib, lsd):
    from deft_amd import integrate, array_tracker as MT, tracker as DT
    MT.TrackIds.count = 0
    afe = FakeAFE()
    afe.plan = types.SimpleNamespace(lib=emu_lib, _stream=lambda: None)
    model = types.SimpleNamespace(AFE=afe)
    if opt.lstm:
        model.motion = DT.MotionBank(integrate.KalmanFilterLSTM(opt, lsd, device="cpu", lib=emu_lib))
    return MT.ArrayTracker(opt, model, h=H, w=W)


@pytest.mark.parametrize("dataset", ["mot", "kitti_tracking"])
def test_array_tracker_lstm_matches_reference_tracker(emu_lib, dataset):
    """`--lstm` on the 2-D datasets (BASELINE configs[3]: KITTI + LSTM motion gating): no Kalman predict, the LSTM's predicted boxes in
    the IoU stage (matching.py:93-96), one deft_motion_step launch per frame for every track touched."""
    import deft_oracle as O
    torch.set_grad_enabled(False)
    lsd = O.synth_lstm_state_dict("mot")
    nframes = 40
    opt, ref, (RT, ref_cls) = _reference_lstm(dataset, types.SimpleNamespace(AFE=FakeAFE()), True, lsd)
    try:
        mine = _mine(opt, emu_lib, lsd)
        fm = [torch.zeros(1, 1, 1, 1)]
        ids = set()
        for t in range(nframes):
            res = _scene(t, dataset)
            a = _log(ref.update([dict(r) for r in res], fm))
            b = _log(mine.update([dict(r) for r in res], fm))
            assert [x[:3] for x in a] == [x[:3] for x in b], (t, a, b)
            for x, y in zip(a, b):
                assert np.abs(np.array(x[3]) - np.array(y[3])).max() <= 1e-9 and x[4] == y[4], (t, x, y)
            ids |= {x[0] for x in a}
            assert sorted(tk.track_id for tk in ref.tracked_stracks) == sorted(tk.track_id for tk in mine.tracked_stracks), t
            # the LSTM side: the predictions the next frame's IoU stage will read (float32 on both sides)
            want = {tk.track_id: tk.future_predictions for tk in ref.tracked_stracks}
            mine._resolve()
            for row, tid in enumerate(mine.cols["tid"].tolist()):
                for k, v in want[tid].items():
                    assert np.abs(mine.fut_arr[row, k - 1] - np.asarray(v, dtype=np.float64)).max() <= 2e-3 * max(1.0, float(np.abs(v).max())), (t, tid, k)
        assert len(ids) >= 5 and mine.bank.launches <= nframes
        assert len(mine.removed_stracks) == len(ref.removed_stracks)
    finally:
        RT.KalmanFilterLSTM = ref_cls
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)


def test_lstm_gate_with_300_observations_matches_fuse_motion(emu_lib):
    """matching.fuse_motion's branch for LSTM tracks with >= 300 observations (matching.py:342-353: Mahalanobis on the predicted box with
    np.cov of the observations): the tracker's running scatter against np.cov, and its gate against deft_amd.association.fuse_motion
    (pinned to the reference by tests/golden/association.npz)."""
    from deft_amd import association as A, array_tracker as MT
    g = np.random.RandomState(4)
    T, n = 6, 320
    obs = g.randn(T, n, 4) * np.array([30, 20, 0.05, 10]) + np.array([300, 200, 0.5, 80])
    opt = types.SimpleNamespace(dataset="mot", track_buffer=30, max_object=100, lstm=True)
    bank = types.SimpleNamespace(alloc=lambda: 0, free=lambda s: None)
    trk = MT.ArrayTracker(opt, types.SimpleNamespace(AFE=None, motion=bank), h=H, w=W)
    trk.cols.append(T, tid=np.arange(1, T + 1))
    for k in range(n):
        trk._observe(np.arange(T), obs[:, k])
    cov = trk.cols["om2"] / (trk.cols["nobs"] - 1)[:, None, None]
    for t in range(T):
        assert np.abs(cov[t] - np.cov(obs[t].T)).max() <= 1e-9 * np.abs(np.cov(obs[t].T)).max()
    pred = (obs.mean(1) + g.randn(T, 4)).astype(np.float32)
    meas = obs.mean(1)[g.permutation(T)] + g.randn(T, 4) * np.array([40, 40, 0.01, 3])
    meas[::2, :2] += 400.0                                              # half of the detections far outside every gate
    cost = g.rand(T, T)
    tracks = [types.SimpleNamespace(observations=[0] * n, covariance=np.cov(obs[t].T), prediction_at_frame=(lambda f, t=t: pred[t])) for t in range(T)]
    dets = [types.SimpleNamespace(to_xyah=(lambda r=meas[j]: r.copy())) for j in range(T)]
    want = A.fuse_motion(None, cost.copy(), tracks, dets, frame_id=9, use_lstm=True)
    gm = A._maha2(pred.astype(np.float64)[:, :2], cov[:, :2, :2], meas[:, :2])
    got = cost.copy()
    got[gm > 5.0 * A.chi2inv95[2]] = np.inf
    got = 0.9 * got + 0.05 * (1 - 0.9) * gm
    assert np.isinf(want).any() and np.isfinite(want).any()
    assert (np.isinf(want) == np.isinf(got)).all() and np.abs(want[np.isfinite(want)] - got[np.isfinite(got)]).max() <= 1e-9


def _scene3d(t, classe):
    """One class's detections of a nuScenes camera frame: objects on the ground plane drifting in x / z with a slow yaw, a fake pinhole
    projection for the 2-D box; #2 is missed in frames 5-6, #4 appears at frame 8, #1 leaves at 15; frame 3 is empty."""
    if t == 3 or t >= 24:
        return [], [], [], [], []
    rows, ddd, depth, org, sub = [], [], [], [], []
    for i in range(6):
        if (i == 2 and t in (5, 6)) or (i == 4 and t < 8) or (i == 1 and t >= 15):
            continue
        ped = classe == "pedestrian"
        h, w, l = (1.7, 0.6, 0.7) if ped else (1.5 + 0.05 * i, 1.8 + 0.03 * i, 4.2 + 0.1 * i)
        x = -8.0 + 3.5 * i + (0.25 if ped else 0.6) * t * (1 if i % 2 else -1) + 0.02 * np.sin(t * 1.7 + i)
        z = 12.0 + 4.0 * i + 0.3 * t + 0.03 * np.cos(t * 1.3 + 2 * i)
        y = 1.0 + 0.01 * np.sin(t + i)
        rot = 0.2 * i + 0.02 * t + 0.005 * np.sin(3 * t + i)
        u, v = 100 + 40 * x / z * 6, 60 + 10.0 / z * 20
        bw, bh = (w + l) * 30 / z, h * 60 / z
        rows.append([u - bw / 2, v - bh / 2, u + bw / 2, v + bh / 2, 0.8 - 0.05 * i + 0.01 * np.cos(t + i), 1.0])
        ddd.append([h + 0.01 * np.sin(t + i), w, l + 0.02 * np.cos(t * 0.7 + i), x, y, z, rot])
        depth.append([z])
        org.append({"frame": t, "obj": i, "kind": "org"})
        sub.append({"frame": t, "obj": i, "kind": "submission"})
    return rows, ddd, depth, org, sub


def _log3(targets):
    return [(int(t.track_id), bool(t.is_activated), int(t.tracklet_len), [float(v) for v in t.tlwh], float(t.score),
             [float(v) for v in np.asarray(t.ddd_bbox)], float(t.depth), t.org_ddd_box, t.ddd_submission, t.classe) for t in targets]


@pytest.mark.parametrize("lstm", [False, True])
@pytest.mark.parametrize("classe", ["car", "pedestrian"])
def test_array_tracker_nuscenes_matches_reference_tracker(emu_lib, classe, lstm):
    """BASELINE configs[4]: the per-class nuScenes tracker -- 3-D IoU first association (all classes but pedestrian; matching.py:107-131
    through deft_iou3d_matrix), embedding association with the 3-D motion gate (the centre distance with the LSTM, the squared
    7-component distance with the plain Kalman filter), similarity-only association, 2-D IoU stage at threshold 0 -- against the
    reference's Tracker.update (tracker.py:738-778, 850-953, 999-1004) on the same arguments, Kalman and LSTM."""
    import deft_oracle as O
    torch.set_grad_enabled(False)
    lsd = O.synth_lstm_state_dict("nuscenes")
    opt, ref, (RT, ref_cls) = _reference_lstm("nuscenes", types.SimpleNamespace(AFE=FakeAFE()), lstm, lsd)
    try:
        mine = _mine(opt, emu_lib, lsd)
        fm = [torch.zeros(1, 1, 1, 1)]
        ids, n3d = set(), 0
        for t in range(30):
            rows, ddd, depth, org, sub = _scene3d(t, classe)
            a = _log3(ref.update([list(r) for r in rows], fm, ddd_boxes=[list(d) for d in ddd], depths_by_class=[list(d) for d in depth],
                                 ddd_org_boxes=list(org), submission=list(sub), classe=classe))
            mine_rows, mine_ddd, mine_depth = [list(r) for r in rows], [list(d) for d in ddd], [list(d) for d in depth]
            if t % 3 == 1:                                               # Detector.run begins every class's device half before the first update()
                mine.begin(mine_rows, fm, ddd_boxes=mine_ddd, depths_by_class=mine_depth)
            elif t % 3 == 2:                                             # ... and a begin() for detections that never come is taken back
                mine.begin([list(r) for r in rows[:1]], fm, ddd_boxes=mine_ddd[:1], depths_by_class=mine_depth[:1])
            b = _log3(mine.update(mine_rows, fm, ddd_boxes=mine_ddd, depths_by_class=mine_depth,
                                  ddd_org_boxes=list(org), submission=list(sub), classe=classe))
            assert mine._begun is None
            assert [x[:3] for x in a] == [x[:3] for x in b], (t, [x[:3] for x in a], [x[:3] for x in b])
            for x, y in zip(a, b):
                assert np.abs(np.array(x[3]) - np.array(y[3])).max() <= 1e-9 and x[4] == y[4], (t, x, y)
                assert x[5] == y[5] and x[6] == y[6] and x[7] is y[7] and x[8] is y[8] and x[9] == y[9] == classe
            ids |= {x[0] for x in a}
            n3d += sum(1 for x in a if x[2] > 0)
            assert [tk.track_id for tk in ref.tracked_stracks] == [tk.track_id for tk in mine.tracked_stracks], t
        assert len(ids) >= 5 and n3d > 40
    finally:
        RT.KalmanFilterLSTM = ref_cls
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)
