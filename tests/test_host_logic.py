"""-m "not gpu": host-side logic of the seam objects that needs no reference checkout: the motion-model
seam's constructor/caching and `gating_distance` (kalman_filter_lstm.py:80-102), node selection of the track
similarity (tracker.py:219-252), slot management of the motion bank."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from deft_amd import integrate, tracker as DT


def _opt(dataset="mot"):
    return SimpleNamespace(dataset=dataset, load_model_traj="", gpus=[-1])


def test_kalman_lstm_seam_constructs_like_the_reference(emu_lib, tmp_path):
    """`KalmanFilterLSTM(opt)` (tracker.py:144, 301, 661): no state dict argument; weights from
    opt.load_model_traj ("module." prefixes stripped, model.py:49-53) or a fresh DecoderRNN; the packed
    plan is shared between instances."""
    a = integrate.KalmanFilterLSTM(_opt(), lib=emu_lib)
    b = integrate.KalmanFilterLSTM(_opt(), lib=emu_lib)
    assert a.plan is b.plan and a.MAX_dis_fut == 5 and a.plan.nin == 11 and a.plan.nout == 20
    n = integrate.KalmanFilterLSTM(_opt("nuscenes"), lib=emu_lib)
    assert n.MAX_dis_fut == 4 and n.plan.nin == 18 and n.plan.nout == 16 and n.plan is not a.plan
    import deft_oracle as O
    lsd = O.synth_lstm_state_dict("mot")
    path = str(tmp_path / "traj.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in lsd.items()}}, path)
    o = _opt(); o.load_model_traj = path
    k = integrate.KalmanFilterLSTM(o, lib=emu_lib)
    x = torch.randn(1, 1, 11, generator=torch.Generator().manual_seed(0))
    h, c, pred = k.predict(torch.zeros(1, 1, 128), torch.zeros(1, 1, 128), x)
    ho, co, po = O.lstm_predict(torch.zeros(1, 128), torch.zeros(1, 128), x[0], lsd)
    assert (h.view(-1) - ho.view(-1)).abs().max() <= 1e-5 and sorted(pred) == [1, 2, 3, 4, 5]
    assert np.abs(np.stack([pred[i] for i in range(1, 6)]) - po[0].numpy()).max() <= 1e-5


def test_gating_distance():
    k = integrate.KalmanFilterLSTM.__new__(integrate.KalmanFilterLSTM)      # host-only method
    mean = np.array([10.0, 20.0, 0.5, 40.0])
    cov = np.diag([4.0, 9.0, 1.0, 1.0])
    meas = np.array([[12.0, 23.0, 0.5, 41.0], [10.0, 20.0, 0.7, 40.0]])
    d = k.gating_distance(mean, cov, meas, only_position=True, metric="maha")
    assert np.allclose(d, [(2 / 2) ** 2 + (3 / 3) ** 2, 0.0])
    d4 = k.gating_distance(mean, cov, meas, only_position=False, metric="maha")
    assert np.allclose(d4, [1 + 1 + 0 + 1, 0.04])
    # the reference's "gaussian" metric on the 2-D path looks at an empty slice: distance 0 for everything
    assert np.array_equal(k.gating_distance(mean, cov, meas, only_position=True, metric="gaussian"), [0.0, 0.0])
    # 3-D (7 components): centre distance over components 3..5
    m7 = np.array([1.5, 1.8, 4.0, 2.0, 1.0, 30.0, 0.1])
    z7 = np.array([[1.4, 1.7, 4.2, 5.0, 1.0, 34.0, 0.3]])
    assert np.allclose(k.gating_distance(m7, np.eye(7), z7, only_position=False, metric="gaussian"), [5.0])
    with pytest.raises(ValueError):
        k.gating_distance(mean, cov, meas, metric="euclid")


def test_select_nodes():
    N = lambda f: SimpleNamespace(frame_index=f, id=0)
    nodes = [N(f) for f in (1, 40, 50, 55, 56, 57, 58)]
    pick = lambda fr, ds: [n.frame_index for n in DT.select_nodes(nodes, fr, ds)]
    assert pick(59, "mot") == [55, 56, 57, 58]                 # 7 usable > mm+1 -> last mm = 4
    assert pick(59, "nuscenes") == [57, 58]
    assert pick(59, "mot") == pick(59, "kitti_tracking")
    assert [n.frame_index for n in DT.select_nodes(nodes[:5], 59, "mot")] == [40, 50, 55, 56]    # frame 1 is 58 frames old -> dropped
    assert [n.frame_index for n in DT.select_nodes(nodes[1:6], 59, "mot")] == [40, 50, 55, 56, 57]   # mm+1 rows: all
    assert DT.select_nodes(nodes[:1], 59, "mot") == []


def test_motion_bank_slots(emu_lib):
    import deft_oracle as O
    from deft_amd import engine
    bank = DT.MotionBank(engine.LstmPlan(O.synth_lstm_state_dict("mot"), "cpu", emu_lib), capacity=2)
    s = [bank.alloc() for _ in range(5)]
    assert sorted(s) == [0, 1, 2, 3, 4] and bank.h.shape[0] == 8
    bank.step(s[:2], np.array([[10.0, 10, 5, 9], [50.0, 20, 6, 12]]), 1)
    assert float(bank.last[s[0], 0]) == 1.0 and float(bank.last[s[0], 1]) == 1.0
    bank.free(s[0])
    again = bank.alloc()
    assert again == s[0] and float(bank.last[again].abs().sum()) == 0.0 and float(bank.h[again].abs().sum()) == 0.0
    with pytest.raises(AssertionError):
        bank.step([s[1], s[1]], np.zeros((2, 4)) + 5.0, 2)       # one row per track and launch


def test_accelerate_binds_and_restores(emu_lib):
    """deft_amd.tracker.accelerate on a stand-in for the reference's utils.tracker module."""
    import types
    import deft_oracle as O
    from deft_amd import association
    matching = types.SimpleNamespace(fuse_motion=1, fuse_motion_ddd=2, linear_assignment=3, bbox_ious=4)

    class Tracker:
        def get_similarity(self):
            return "ref"

    class STrack:
        def update_lstm_features(self, tlwh):
            return "ref"

        def update_lstm_features_ddd(self, b):
            return "ref"
    RT = types.SimpleNamespace(Tracker=Tracker, FeatureRecorder=object, STrack=STrack, matching=matching, KalmanFilterLSTM=dict)
    ref_get = Tracker.get_similarity
    kf = integrate.KalmanFilterLSTM(_opt(), O.synth_lstm_state_dict("mot"), device="cpu", lib=emu_lib)
    undo = DT.accelerate(RT, kf)
    assert Tracker.get_similarity is DT.get_similarity and RT.FeatureRecorder is DT.FeatureRecorder
    assert matching.fuse_motion is association.fuse_motion and matching.bbox_ious is association.bbox_overlaps
    assert isinstance(STrack.__dict__["future_predictions"], property) and RT.KalmanFilterLSTM is integrate.KalmanFilterLSTM
    undo()
    assert RT.KalmanFilterLSTM is dict
    assert Tracker.get_similarity is ref_get and RT.FeatureRecorder is object and matching.linear_assignment == 3
    assert "future_predictions" not in STrack.__dict__ and STrack().update_lstm_features(None) == "ref"


def test_embed_group_cache_keeps_captured_entries():
    """AfePlan._embed_group: an LRU of EGROUP_CACHE shapes, except that entries a hipGraph replays into (marked while capturing) are never
    dropped -- their buffers are baked into the graph."""
    import types
    import torch
    from deft_amd import engine
    afe = engine.AfePlan.__new__(engine.AfePlan)
    afe.device = torch.device("cpu")
    w = torch.zeros(8, 32 * 9)
    afe.sel = [(w, 32 * 9, torch.zeros(8), 8, 32, 0)]
    fm = [types.SimpleNamespace(addr=4096, H=8, W=8, C=32, N=1, ld=32)]
    first = afe._embed_group(fm, 1, 1)
    first["pinned"] = True                                   # what a hit during capture sets
    for nd in range(2, 2 + 3 * engine.AfePlan.EGROUP_CACHE):
        afe._embed_group(fm, 1, nd)
    keys = list(afe._egroups)
    assert keys[0][2] == 1 and afe._egroups[keys[0]] is first
    assert len(keys) == engine.AfePlan.EGROUP_CACHE + 1
    assert [k[2] for k in keys[1:]] == list(range(2 + 2 * engine.AfePlan.EGROUP_CACHE, 2 + 3 * engine.AfePlan.EGROUP_CACHE))


# ---- bench.py's parity gate: what "decidable" means, checked on the oracle's own decode (no GPU) ----
def _decode(logit, K):
    import deft_oracle as O
    C, h, w = logit.shape
    z = torch.zeros(1, 2, h, w)
    od = O.generic_decode(O.sigmoid_output({"hm": logit[None].clone(), "reg": z, "wh": z}), K=K)
    return (od["clses"][0].long() * h * w + od["inds"][0].long()).tolist()


def test_gate_margin_guarantees_the_ordered_topk():
    """bench.oracle_margin: every heat map within margin / 2 of the oracle's decodes to the same ordered (class, index) list; and a map further
    away than that CAN decode differently, but then only inside bench.tie_class_ok's class."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    g = torch.Generator().manual_seed(0)
    flips = 0
    for t in range(24):
        C = (1, 3, 10)[t % 3]
        logit = torch.randn(C, 20, 28, generator=g) - 2
        K = 15
        m = bench.oracle_margin(logit, K)
        ok = _decode(logit, K)
        for r in range(4):
            pert = logit + (torch.rand(logit.shape, generator=g) * 2 - 1) * (m / 2 * 0.999)
            assert _decode(pert, K) == ok
        e = 0.02                                             # a cross-implementation error far above the margin: lists differ, inside the tie class
        pert = logit + (torch.rand(logit.shape, generator=g) * 2 - 1) * e
        gk = _decode(pert, K)
        flips += gk != ok
        assert bench.tie_class_ok(logit, gk, ok, 2 * e)
    assert flips >= 3                                        # (the class check was exercised on real differences)


def test_gate_tie_class_rejects_a_wrong_list():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    g = torch.Generator().manual_seed(1)
    logit = torch.randn(1, 20, 28, generator=g) - 2
    K = 15
    ok = _decode(logit, K)
    assert bench.tie_class_ok(logit, ok, ok, 1e-6)
    sw = list(ok); sw[0], sw[5] = sw[5], sw[0]               # two clearly different scores swapped
    assert not bench.tie_class_ok(logit, sw, ok, 1e-4)
    flat = logit.reshape(-1)
    low = int(torch.argmin(flat))                            # a pixel that is nowhere near the top K
    bad = list(ok); bad[-1] = low
    assert not bench.tie_class_ok(logit, bad, ok, 1e-4)
    assert not bench.tie_class_ok(logit, ok[:-1] + [ok[0]], ok, 1e-4)      # a repeated key


def test_gate_tie_class_accepts_a_merged_peak():
    """Two oracle peaks A, B with the pixel C between them a hair below both: an implementation whose logits are 1e-4 away may find C on top,
    report ONE peak where the oracle has two, and take the oracle's (K + 1)-th detection in -- however far below the K-th it scores.  That list is
    inside the tie class; a list that takes some other low pixel in is not."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    g = torch.Generator().manual_seed(4)
    logit = torch.full((1, 12, 40), -9.0)
    vals = torch.linspace(3.0, 1.0, 12)
    cols = list(range(2, 38, 3))                               # 12 isolated peaks on row 2 ...
    for v, c in zip(vals, cols):
        logit[0, 2, c] = v
    logit[0, 8, 10], logit[0, 8, 11], logit[0, 8, 12] = 5.0, 5.0 - 4e-5, 5.0 - 2e-5      # ... and A, C, B on row 8
    K = 10
    ok = _decode(logit, K)
    A, Cc, B = 8 * 40 + 10, 8 * 40 + 11, 8 * 40 + 12
    assert ok[0] == A and ok[1] == B                           # the oracle keeps A and B (+ the 8 best of row 2)
    dev = logit.clone(); dev[0, 8, 11] += 1e-4                 # the device sees C on top: one peak instead of two
    gk = _decode(dev, K)
    assert Cc in gk and A not in gk and B not in gk and gk[-1] == 2 * 40 + cols[8]        # the oracle's 11-th detection moved in, 0.18 below the 10-th
    assert bench.tie_class_ok(logit, gk, ok, 2.5e-4)
    bad = list(gk); bad[-1] = 2 * 40 + cols[10]                # a list that skips the oracle's next detection for a lower one
    assert not bench.tie_class_ok(logit, bad, ok, 2.5e-4)


def test_peaked_head_classes():
    """deft_amd.synth.peaked_head with several classes: every class map is a blob map, the K best peaks are well separated (what bench.py's peaked
    gate stream relies on)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    import deft_oracle as O
    from deft_amd.synth import peaked_head
    for ds, H, W in (("kitti_tracking", 96, 320), ("nuscenes", 112, 200), ("mot", 128, 160)):
        C = O.HEADS[ds]["hm"]
        feat, sd2 = peaked_head(O.synth_state_dict(ds), H * 2, W * 2, 60, seed=12, classes=C)
        with torch.no_grad():
            out = {"hm": O.head_forward(feat, sd2, "hm")}
        assert out["hm"].shape[1] == C
        assert bench.oracle_margin(out["hm"][0], 40) >= 1e-3


def test_input_geometry_modes():
    """preprocess.input_geometry: the three input modes of Detector._transform_scale (detector.py:346-376).  Sizes by hand; and, where the
    reference tree is present, (c, s, inp size, trans_input) against the reference's own _transform_scale + get_affine_transform."""
    from deft_amd import preprocess as PR
    o = SimpleNamespace(fix_short=0, fix_res=True, input_h=608, input_w=1088, pad=31)
    M, c, s, ih, iw = PR.input_geometry(o, 1080, 1920)
    assert (ih, iw) == (608, 1088) and float(s) == 1920.0 and c.tolist() == [960.0, 540.0]
    M, c, s, ih, iw = PR.input_geometry(SimpleNamespace(fix_short=512, fix_res=True, input_h=608, input_w=1088, pad=31), 1080, 1920)
    assert (ih, iw) == (512, 960) and s.tolist() == [1920.0, 1080.0]            # int(1920 / 1080 * 512) = 910 -> 960
    M, c, s, ih, iw = PR.input_geometry(SimpleNamespace(fix_short=0, fix_res=False, input_h=0, input_w=0, pad=31), 375, 1242)
    assert (ih, iw) == (384, 1248) and s.tolist() == [1248.0, 384.0] and c.tolist() == [621.0, 187.0]
    if not os.path.isdir("/root/reference/src/lib"):
        return
    import ref_import, ref_shims, make_golden as MG
    ref_shims.install(); ref_import.install_stubs(MG.OracleDCN); ref_shims.install_detector_stubs()
    import cv2
    cv2.resize = lambda im, size: im                                        # (scale 1: the identity)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from detector import Detector as RefDetector
        from utils.image import get_affine_transform
    finally:
        sys.argv = argv
    for h, w in ((1080, 1920), (375, 1242), (900, 1600), (480, 640), (640, 480)):
        for o in (SimpleNamespace(fix_short=0, fix_res=True, input_h=608, input_w=1088, pad=31), SimpleNamespace(fix_short=512, fix_res=True, input_h=0, input_w=0, pad=31),
                  SimpleNamespace(fix_short=0, fix_res=False, input_h=0, input_w=0, pad=31)):
            img = np.zeros((h, w, 3), np.uint8)
            _, rc, rs, riw, rih, _, _ = RefDetector._transform_scale(SimpleNamespace(opt=o), img)
            rM = get_affine_transform(rc, rs, 0, [riw, rih])
            M, c, s, ih, iw = PR.input_geometry(o, h, w)
            assert (ih, iw) == (rih, riw) and np.array_equal(np.asarray(c), np.asarray(rc)) and np.array_equal(np.asarray(s), np.asarray(rs))
            assert np.abs(M - rM).max() <= 1e-12
