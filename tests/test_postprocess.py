"""-m "not gpu": deft_amd/postprocess.py against fixtures written from the reference's own
`generic_post_process` / `nms` (tests/golden/postprocess.npz, oracle/make_golden.py run_postprocess), and -- for the parts whose
third-party dependencies (pyquaternion, nuscenes Box) exist nowhere here: PARITY UNPINNED -- against an independent scalar
restatement on scipy.spatial.transform.Rotation."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

from deft_amd import postprocess as PP

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.npz")


@pytest.mark.parametrize("tag,ddd", [("mot", False), ("nusc", True)])
def test_generic_post_process_matches_reference(tag, ddd):
    f = np.load(GOLD)
    dets = {k[len(tag) + 4:]: f[k] for k in f.files if k.startswith(tag + "_in_")}
    oh, ow = [int(v) for v in f[tag + "_hw"]]
    post = PP.generic_post_process(dets, f[tag + "_c"], f[tag + "_s"], oh, ow, 0.25, calib=f[tag + "_calib"])
    n = int(f[tag + "_n"])
    assert post["score"].shape[0] == n                                  # the reference's `break` at the first score < out_thresh
    assert np.array_equal(post["class"], f[tag + "_out_class"].reshape(-1).astype(np.int64))
    for key in ("score", "ct", "bbox", "tracking") + (("dep", "dim", "alpha", "loc", "rot_y") if ddd else ()):
        ref = f["%s_out_%s" % (tag, key)].reshape(np.asarray(post[key]).shape)
        err = float(np.abs(np.asarray(post[key], np.float64) - ref).max() / max(1.0, np.abs(ref).max()))
        assert err <= 1e-6, (key, err)
    merged = PP.merge_outputs(post, 0.25)
    assert merged["score"].shape[0] == int((f[tag + "_out_score"].reshape(-1) > 0.25).sum())
    lst = PP.as_result_list(merged)
    assert isinstance(lst[0]["class"], int) and lst[0]["bbox"].shape == (4,)


def test_greedy_nms_matches_reference_incl_the_keep_quirk():
    f = np.load(GOLD)
    for case in range(4):
        keep, count = PP.greedy_nms(f["nms%d_boxes" % case], f["nms%d_scores" % case], overlap=float(f["nms%d_overlap" % case]))
        assert count == int(f["nms%d_count" % case]) and np.array_equal(keep, f["nms%d_keep" % case])
        if count < keep.shape[0]:
            assert 0 in set(keep.tolist())                              # zero padding: `sorted(set(keep))` always contains index 0
    keep, count = PP.greedy_nms(np.zeros((0, 4)), np.zeros(0))
    assert keep.shape == (0,) and count == 0


def test_native_greedy_nms_equals_numpy_and_reference_vectors(emu_lib):
    """deft_greedy_nms (assoc.hip, one host call per class) against the reference's golden vectors and against the numpy loop on random crowded
    scenes: equal scores (stable order), boxes of zero area (a NaN ratio drops the box in both), more boxes than top_k."""
    f = np.load(GOLD)
    for case in range(4):
        keep, count = PP.greedy_nms(f["nms%d_boxes" % case], f["nms%d_scores" % case], overlap=float(f["nms%d_overlap" % case]), lib=emu_lib)
        assert count == int(f["nms%d_count" % case]) and np.array_equal(keep, f["nms%d_keep" % case])
    g = np.random.RandomState(9)
    for trial in range(40):
        n = int(g.randint(1, 60))
        xy = g.rand(n, 2) * 50
        wh = g.rand(n, 2) * 30 * (g.rand(n, 1) > 0.1)                   # a tenth of the boxes degenerate
        boxes = np.concatenate([xy, xy + wh], 1)
        scores = np.round(g.rand(n), 1 if trial % 2 else 6)              # every other trial: many equal scores
        for top_k in (200, 7):
            a = PP.greedy_nms(boxes, scores, overlap=0.3 + 0.1 * (trial % 5), top_k=top_k, lib=emu_lib)
            b = PP.greedy_nms(boxes, scores, overlap=0.3 + 0.1 * (trial % 5), top_k=top_k)
            assert a[1] == b[1] and np.array_equal(a[0], b[0]), (trial, top_k)
    keep, count = PP.greedy_nms(np.zeros((0, 4)), np.zeros(0), lib=emu_lib)
    assert keep.shape == (0,) and count == 0


def test_quaternion_helpers_against_scipy():
    g = np.random.RandomState(0)
    a = g.randn(20, 4); b = g.randn(20, 4)
    a /= np.linalg.norm(a, axis=1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
    sa, sb = R.from_quat(a[:, [1, 2, 3, 0]]), R.from_quat(b[:, [1, 2, 3, 0]])
    assert np.abs(PP.q_matrix(a) - sa.as_matrix()).max() <= 1e-12
    prod = PP.q_mul(a, b)
    assert np.abs(PP.q_matrix(prod) - (sa * sb).as_matrix()).max() <= 1e-12
    th = g.randn(20) * 2
    assert np.abs(PP.q_matrix(PP.q_axis_angle([0, 1, 0], th)) - R.from_rotvec(np.outer(th, [0, 1, 0])).as_matrix()).max() <= 1e-12
    ang, axis = PP.q_angle_axis(a)
    assert np.all(ang > -np.pi) and np.all(ang <= np.pi)
    assert np.abs(PP.q_matrix(PP.q_axis_angle(axis[3], ang[3:4])) - sa[3:4].as_matrix()).max() <= 1e-9   # (angle, axis) reproduce the rotation


def _scalar_nuscenes(post, info):
    """The reference's per-detection loop (detector.py:200-300), restated with scipy rotations (independent of PP's quaternion code)."""
    out = {n: {"results": [], "ddd_boxes": [], "depths": [], "ddd_org_boxes": [], "submission": []} for n in PP.NUSCENES_TRACKING_NAMES}
    tm = np.array(info["trans_matrix"], np.float32)
    for i in range(post["score"].shape[0]):
        name = PP.NUSCENES_CLASS_NAME[int(post["class"][i]) - 1]
        sc = post["score"][i]
        if name not in PP.NUSCENES_TRACKING_NAMES or sc < 0.3 or (name == "pedestrian" and sc < 0.35):
            continue
        dim, loc, ry = post["dim"][i], post["loc"][i], float(post["rot_y"][i])
        size = [float(dim[1]), float(dim[2]), float(dim[0])]
        t1 = np.dot(tm, np.array([loc[0], loc[1] - size[2], loc[2], 1], np.float32))
        rot = R.from_rotvec([0.0, ry, 0.0])
        center = np.array(loc, np.float64) + np.array([0, -size[2] / 2, 0])
        for q, t in ((info["cs_record_rot"], info["cs_record_trans"]), (info["pose_record_rot"], info["pose_record_trans"])):
            rq = R.from_quat([q[1], q[2], q[3], q[0]])
            center = rq.apply(center) + np.asarray(t, np.float64)
            rot = rq * rot
        qx, qy, qz, qw = rot.as_quat()
        if qw < 0:                                   # scipy may return -q; pyquaternion keeps the product's sign: compare up to sign below
            pass
        vn = np.linalg.norm([qx, qy, qz])
        ang = ((2 * np.arctan2(vn, qw) + np.pi) % (2 * np.pi)) - np.pi
        angle = ang if qz / vn > 0 else -ang
        o = out[name]
        o["results"].append(list(post["bbox"][i]) + [sc]); o["depths"].append([float(loc[2])])
        o["ddd_boxes"].append([size[2], size[0], size[1], center[0], center[1], center[2], angle])
        o["ddd_org_boxes"].append([float(dim[0]), float(dim[1]), float(dim[2]), loc[0], loc[1], loc[2], ry])
        o["submission"].append([float(t1[0]), float(t1[1]), float(t1[2])] + size + [qw, qx, qy, qz])
    return out


def test_nuscenes_branch_against_scalar_restatement():
    f = np.load(GOLD)
    dets = {k[8:]: f[k] for k in f.files if k.startswith("nusc_in_")}
    dets["scores"] = np.clip(dets["scores"] + 0.15, 0, 1)                   # more rows above the 0.3 / 0.35 class thresholds
    oh, ow = [int(v) for v in f["nusc_hw"]]
    post = PP.merge_outputs(PP.generic_post_process(dets, f["nusc_c"], f["nusc_s"], oh, ow, 0.1, calib=f["nusc_calib"]), 0.1)
    g = np.random.RandomState(3)
    q1, q2 = g.randn(4), g.randn(4)
    info = {"trans_matrix": np.concatenate([R.from_rotvec(g.randn(3)).as_matrix(), g.randn(3, 1) * 10], 1),
            "cs_record_rot": (q1 / np.linalg.norm(q1)).tolist(), "cs_record_trans": [1.7, 0.0, 1.5],
            "pose_record_rot": (q2 / np.linalg.norm(q2)).tolist(), "pose_record_trans": [411.3, 1180.9, 0.0]}
    got = PP.nuscenes_frame(post, info, nms=False)
    ref = _scalar_nuscenes(post, info)
    nrows = 0
    for name in PP.NUSCENES_TRACKING_NAMES:
        for key in ("results", "ddd_boxes", "depths", "ddd_org_boxes"):
            a, b = np.asarray(got[name][key], np.float64), np.asarray(ref[name][key], np.float64).reshape(np.asarray(got[name][key]).shape)
            assert a.shape == b.shape
            if a.size:
                assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), (name, key)
        a, b = np.asarray(got[name]["submission"]), np.asarray(ref[name]["submission"], np.float64).reshape(np.asarray(got[name]["submission"]).shape)
        if a.size:
            assert np.abs(a[:, :6] - b[:, :6]).max() <= 1e-4
            sign = np.sign(np.sum(a[:, 6:] * b[:, 6:], axis=1, keepdims=True))       # q and -q are the same rotation
            assert np.abs(a[:, 6:] - sign * b[:, 6:]).max() <= 1e-9
        nrows += a.shape[0]
    assert nrows >= 8
    # with NMS: per class, the kept set is sorted(set(keep)) of greedy_nms -- index 0 of the class always survives
    got_nms = PP.nuscenes_frame(post, info, nms=True)
    for name in PP.NUSCENES_TRACKING_NAMES:
        full, kept = got[name]["results"], got_nms[name]["results"]
        if len(full):
            keep, _ = PP.greedy_nms(full[:, :4], full[:, -1], overlap=0.7 if name in ("bus", "truck") else 0.8)
            assert np.array_equal(kept, full[sorted(set(keep.tolist()))]) and np.array_equal(kept[0], full[0])


def test_detector_mirror_post_process_methods():
    """deft_amd.detector.Detector.post_process / merge_outputs / nuscenes_targets: the reference's method names and return
    shapes (detector.py:553-583, 200-338) on top of the vectorised functions -- checked against the same reference fixtures."""
    from types import SimpleNamespace
    from deft_amd.detector import Detector
    f = np.load(GOLD)
    det = Detector.__new__(Detector)                     # the methods under test touch no device state
    det.opt = SimpleNamespace(out_thresh=0.25)
    dets = {k[8:]: f[k] for k in f.files if k.startswith("nusc_in_")}
    oh, ow = [int(v) for v in f["nusc_hw"]]
    meta = {"c": f["nusc_c"], "s": f["nusc_s"], "out_height": oh, "out_width": ow, "calib": f["nusc_calib"]}
    res = det.post_process(dets, meta)
    assert len(res) == int(f["nusc_n"]) and abs(float(res[3]["rot_y"]) - float(f["nusc_out_rot_y"][3])) <= 1e-6
    merged = det.merge_outputs([res])
    assert len(merged) == int((f["nusc_out_score"].reshape(-1) > 0.25).sum())
    info = {"trans_matrix": np.eye(3, 4), "cs_record_rot": [1, 0, 0, 0], "cs_record_trans": [0, 0, 0], "pose_record_rot": [1, 0, 0, 0], "pose_record_trans": [0, 0, 0]}
    by_class = det.nuscenes_targets(merged, info)
    assert hasattr(merged, "arrays") and merged.arrays() is not None       # the arrays rode along: no parsing back ...
    slow = det.nuscenes_targets(list(merged), info)                        # ... and the parsed form gives the same per-class arguments
    for name in by_class:
        for k in by_class[name]:
            assert np.array_equal(np.asarray(by_class[name][k]), np.asarray(slow[name][k])), (name, k)
    assert set(by_class) == set(__import__("deft_amd.postprocess", fromlist=["x"]).NUSCENES_TRACKING_NAMES)
    assert sum(len(v["results"]) for v in by_class.values()) > 0
    assert det.nuscenes_targets([], info)["car"]["results"] == []


def test_result_list_arrays_equal_the_parsed_dicts():
    """postprocess.ResultList: Detector.post_process' list of dicts carries the arrays it was built from, and ArrayTracker.detections_as_arrays
    reads those -- the same rows as parsing the dicts back (the reference's own loop, tracker.py:786-797), through merge_outputs' score cut, for
    the class filter of KITTI, and NOT once somebody has edited the list."""
    from types import SimpleNamespace
    from deft_amd.detector import Detector
    import torch
    from deft_amd import array_tracker as MT
    g = np.random.RandomState(4)
    n = 40
    post = {"score": np.sort(g.rand(n).astype(np.float32))[::-1].copy(), "class": g.randint(1, 4, n).astype(np.int64),
            "ct": g.rand(n, 2).astype(np.float32) * 100, "bbox": (g.rand(n, 4) * 300).astype(np.float32), "tracking": g.randn(n, 2).astype(np.float32)}
    det = Detector.__new__(Detector)
    det.opt = SimpleNamespace(out_thresh=0.3)
    res = PP.as_result_list(post)
    merged = det.merge_outputs([res])
    assert isinstance(merged, PP.ResultList) and len(merged) == int((post["score"] > 0.3).sum()) < n and merged.arrays() is not None
    assert all(a is b for a, b in zip(merged, res))                       # the same dict objects as the plain filter keeps
    for dataset in ("mot", "kitti_tracking"):
        trk = MT.ArrayTracker.__new__(MT.ArrayTracker)
        trk.ddd, trk.dataset, trk.img_width, trk.img_height = False, dataset, 640.0, 480.0
        fast = trk.detections_as_arrays(merged)
        slow = trk.detections_as_arrays(list(merged))
        assert fast["nd0"] == slow["nd0"] > 0
        for k in ("tlwh", "xyah", "tlbr", "dscore", "org"):
            assert np.array_equal(fast[k], slow[k]) and fast[k].dtype == slow[k].dtype, (dataset, k)
        assert torch.equal(fast["centers"], slow["centers"])
    merged.pop()                                                          # an edited list no longer matches its arrays: parsed again
    assert merged.arrays() is None
    assert trk.detections_as_arrays(merged)["nd0"] == trk.detections_as_arrays(list(merged))["nd0"]
    empty = det.merge_outputs([PP.as_result_list({k: v[:0] for k, v in post.items()})])
    assert trk.detections_as_arrays(empty)["nd0"] == 0
