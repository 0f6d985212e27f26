"""-m "not gpu" (build container only: needs /root/reference).  The literal drop-in of BASELINE.json's north star:
`from detector import Detector` as src/test.py:19 does it, resolved to THIS repository's detector.py (a subclass of the
reference's Detector), against the reference's own Detector -- same checkpoint file, same pre-processed frames through
`Detector.run(...)` (pre-processed branch, test.py:213), including the reference's post-processing and its Tracker.  The HIP
kernels run through the SIMT emulator (DEFT_HIP_LIB)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

HAVE_REF = os.path.isdir("/root/reference/src/lib")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.skipif(not HAVE_REF, reason="/root/reference only exists in the build container"), pytest.mark.slow]


def _frame(seed, H, W):
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(seed))
    c = np.array([W / 2.0, H / 2.0], dtype=np.float32)
    s = float(max(H, W))
    meta = {"c": c, "s": np.float32(s), "height": H, "width": W, "out_height": H // 4, "out_width": W // 4,
            "inp_height": H, "inp_width": W, "calib": np.eye(3, 4, dtype=np.float32)}
    batch = lambda v: torch.from_numpy(np.asarray(v)[None])                  # the DataLoader's batch dimension (test.py:106-112)
    return {"image": [torch.zeros(H, W, 3)], "images": {1.0: [x]}, "meta": {1.0: {k: batch(v) for k, v in meta.items()}}}


def test_detector_shim_matches_reference_detector(emu_lib, tmp_path, monkeypatch):
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    ref_shims.install_detector_stubs()
    monkeypatch.setenv("DEFT_HIP_LIB", emu_lib.path)
    from deft_amd import hiplib
    monkeypatch.setattr(hiplib, "_lib", emu_lib)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)       # detector.py:188, 534 call it unconditionally
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from dataset.dataset_factory import dataset_factory
        from utils.basetrack import BaseTrack
        spec = importlib.util.spec_from_file_location("detector", os.path.join(ROOT, "detector.py"))   # what `import detector` finds first
        shim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shim)
    finally:
        sys.argv = argv
    RD = sys.modules["deft_reference_detector"]
    assert issubclass(shim.Detector, RD.Detector) and shim.NUSCENES_TRACKING_NAMES is RD.NUSCENES_TRACKING_NAMES

    sd = dict(O.synth_state_dict("mot"))
    # random regression heads give boxes with negative extent (the tracker's Kalman filter then fails on both sides):
    # bias the amodal l/t/r/b head so that boxes are ~10 x 16 map pixels around the centre
    sd["ltrb_amodal.2.weight"] = sd["ltrb_amodal.2.weight"] * 0.05
    sd["ltrb_amodal.2.bias"] = torch.tensor([-5.0, -8.0, 5.0, 8.0])
    ck = str(tmp_path / "model_mot.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}}, ck)
    H, W = 64, 96
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1", "--load_model", ck, "--K", "8", "--ltrb_amodal",
                        "--input_h", str(H), "--input_w", str(W)])
    opt = opts().update_dataset_info_and_set_heads(opt, dataset_factory[opt.test_dataset])
    opt.out_thresh = 0.0                                         # random-weight scores are 0.01 .. 0.2: keep the K detections
    torch.set_grad_enabled(False)
    try:
        def run(cls):
            BaseTrack._count = 0
            det = cls(opt)
            try:
                det.reset_tracking(opt)                         # test.py:199, once per video
                det.img_height, det.img_width = H, W             # test.py:163-164
                log = []
                for t in range(3):
                    targets = det.run(_frame(10 + t, H, W), image_info={})
                    log.append(sorted((s.track_id, [float(v) for v in s.tlwh], float(s.score)) for s in targets))
            finally:
                if hasattr(det, "_undo_tracker"):
                    det._undo_tracker()                         # leave the reference's tracker module as found (other tests)
            return log

        def run_fused():
            """deft_amd.detector.Detector.run -- process, the vectorised post_process / merge_outputs and the tracker hand-over composed
            in this repository (not inherited from the reference) -- driving the reference's own Tracker with the device forms bound."""
            from deft_amd import checkpoint, detector as FD, integrate, tracker as DT
            import utils.tracker as RT
            BaseTrack._count = 0
            undo = DT.accelerate(RT, None)
            try:
                sdl = checkpoint.load_model_state(ck, opt, log=lambda *_: None)
                fd = FD.Detector(opt, sdl)
                model = integrate.create_model(opt, sdl, device="cpu")
                fd.set_tracker(RT.Tracker(opt, model, h=fd.img_height, w=fd.img_width))      # = reset_tracking BEFORE the sizes are set (100 x 100), as `run` above
                fd.img_height, fd.img_width = H, W
                log = []
                for t in range(3):
                    targets = fd.run(_frame(10 + t, H, W), image_info={})
                    log.append(sorted((s.track_id, [float(v) for v in s.tlwh], float(s.score)) for s in targets))
                assert set(fd.times) == {"load", "pre", "net", "dec", "post", "merge", "track", "tot"}
            finally:
                undo()
            return log

        ref = run(RD.Detector)
        got = run(shim.Detector)
        fused = run_fused()
        assert sum(len(f) for f in ref) >= 8
        for other in (got, fused):
            for fa, fb in zip(ref, other):
                assert [a[0] for a in fa] == [b[0] for b in fb]
                for a, b in zip(fa, fb):
                    assert np.abs(np.array(a[1]) - np.array(b[1])).max() <= 1e-3 and abs(a[2] - b[2]) <= 1e-4
    finally:
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)


def _nusc_info():
    from scipy.spatial.transform import Rotation as R
    g = np.random.RandomState(3)
    q1, q2 = g.randn(4), g.randn(4)
    return {"trans_matrix": np.concatenate([R.from_rotvec(g.randn(3)).as_matrix(), g.randn(3, 1) * 10], 1).tolist(),
            "cs_record_rot": (q1 / np.linalg.norm(q1)).tolist(), "cs_record_trans": [1.7, 0.0, 1.5],
            "pose_record_rot": (q2 / np.linalg.norm(q2)).tolist(), "pose_record_trans": [411.3, 1180.9, 0.0]}


def nusc_state_dict(O):
    """Synthetic nuScenes net whose detections survive the 0.3 / 0.35 class thresholds of detector.py:222-225 and have positive sizes."""
    sd = dict(O.synth_state_dict("nuscenes"))
    sd["hm.2.weight"] = sd["hm.2.weight"] * 3.0
    sd["hm.2.bias"] = torch.tensor([-1.0, -0.8, -1.2, -0.9, -1.0, -1.1, -0.7, -1.0, -1.0, -1.0])
    sd["dim.2.weight"] = sd["dim.2.weight"] * 0.05
    sd["dim.2.bias"] = torch.tensor([1.6, 1.7, 4.0])
    sd["wh.2.weight"] = sd["wh.2.weight"] * 0.05
    sd["wh.2.bias"] = torch.tensor([6.0, 5.0])
    return sd


def _nusc_frame(seed, H, W):
    f = _frame(seed, H, W)
    f["meta"][1.0]["calib"] = torch.from_numpy(np.array([[[60.0, 0, W / 2, 0], [0, 60.0, H / 2, 0], [0, 0, 1, 0]]], np.float32))
    return f


def _target_row(s):
    return (s.track_id, s.classe, [float(v) for v in s.tlwh], float(s.score), [float(v) for v in s.ddd_bbox], [float(v) for v in s.ddd_submission])


@pytest.mark.parametrize("lstm", [False, True])
def test_nuscenes_run_matches_reference_detector(emu_lib, tmp_path, monkeypatch, lstm):
    """BASELINE configs[4] at `online_targets` level: the reference's OWN nuScenes branch of Detector.run (detector.py:200-338: class
    thresholds, pyquaternion / nuscenes `Box` chain, per-class NMS, seven per-class Trackers with the 3-D association of tracker.py:836-960)
    against (a) this repository's detector.py shim behind the same loop and (b) the fused deft_amd.detector.Detector.run (vectorised
    post-processing + `postprocess.nuscenes_frame`) feeding the reference's Trackers.  pyquaternion and the nuScenes devkit are absent:
    the reference side runs on tests/ref_shims.{Quaternion, Box} (PARITY UNPINNED, scipy rotations)."""
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    ref_shims.install_detector_stubs()
    monkeypatch.setenv("DEFT_HIP_LIB", emu_lib.path)
    from deft_amd import hiplib
    monkeypatch.setattr(hiplib, "_lib", emu_lib)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    argv, sys.argv = sys.argv, ["test.py", "tracking,ddd"]
    try:
        from opts import opts
        from dataset.dataset_factory import dataset_factory
        from utils.basetrack import BaseTrack
        import utils.tracker as RT
        spec = importlib.util.spec_from_file_location("detector", os.path.join(ROOT, "detector.py"))
        shim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shim)
    finally:
        sys.argv = argv
    RD = sys.modules["deft_reference_detector"]
    assert RD.Quaternion is ref_shims.Quaternion and RD.Box is ref_shims.Box
    ck = str(tmp_path / "model_nusc.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in nusc_state_dict(O).items()}}, ck)
    H, W, T = 64, 96, 3
    opt = opts().parse(["tracking,ddd", "--dataset", "nuscenes", "--gpus", "-1", "--load_model", ck, "--K", "12",
                        "--input_h", str(H), "--input_w", str(W)])
    opt = opts().update_dataset_info_and_set_heads(opt, dataset_factory[opt.test_dataset])
    opt.lstm = bool(lstm)
    info = _nusc_info()
    lsd = O.synth_lstm_state_dict("nuscenes")
    KF0 = RT.KalmanFilterLSTM

    class KF(KF0):                                              # the reference loads opt.load_model_traj from disk: synthetic weights instead
        def __init__(self, o):
            super().__init__(o)
            self.model.load_state_dict(lsd, strict=True); self.model.eval()
    torch.set_grad_enabled(False)
    try:
        if lstm:
            monkeypatch.setattr(RT, "KalmanFilterLSTM", KF)
            monkeypatch.setattr(RT.STrack, "shared_kalman_lstm", KF(opt))

        def run(cls):
            BaseTrack._count = 0
            det = cls(opt)
            try:
                det.reset_tracking(opt)
                det.img_height, det.img_width = H, W
                return [sorted(_target_row(s) for s in det.run(_nusc_frame(10 + t, H, W), image_info=info)) for t in range(T)]
            finally:
                if hasattr(det, "_undo_tracker"):
                    det._undo_tracker()

        def run_fused():
            from deft_amd import checkpoint, detector as FD, integrate, tracker as DT
            BaseTrack._count = 0
            kf = integrate.KalmanFilterLSTM(opt, lsd, lib=emu_lib) if lstm else None
            undo = DT.accelerate(RT, kf)
            try:
                sdl = checkpoint.load_model_state(ck, opt, log=lambda *_: None)
                fd = FD.Detector(opt, sdl)
                model = integrate.create_model(opt, sdl, device="cpu")
                fd.set_tracker({n: RT.Tracker(opt, model, h=fd.img_height, w=fd.img_width) for n in RD.NUSCENES_TRACKING_NAMES})
                fd.img_height, fd.img_width = H, W
                return [sorted(_target_row(s) for s in fd.run(_nusc_frame(10 + t, H, W), image_info=info)) for t in range(T)]
            finally:
                undo()

        ref = run(RD.Detector)
        assert sum(len(f) for f in ref) >= 10 and len({r[1] for f in ref for r in f}) >= 2
        others = [("fused", run_fused())] + ([] if lstm else [("shim", run(shim.Detector))])      # (the shim once: CPU minutes)
        for name, other in others:
            for t, (fa, fb) in enumerate(zip(ref, other)):
                assert [(a[0], a[1]) for a in fa] == [(b[0], b[1]) for b in fb], (name, t)
                for a, b in zip(fa, fb):
                    assert np.abs(np.array(a[2]) - np.array(b[2])).max() <= 1e-3 and abs(a[3] - b[3]) <= 1e-4, (name, t)
                    assert np.abs(np.array(a[4]) - np.array(b[4])).max() <= 1e-3 * max(1.0, np.abs(np.array(a[4])).max()), (name, t)
                    qa, qb = np.array(a[5]), np.array(b[5])
                    assert np.abs(qa[:6] - qb[:6]).max() <= 1e-3 * max(1.0, np.abs(qa[:6]).max()), (name, t)
                    assert min(np.abs(qa[6:] - qb[6:]).max(), np.abs(qa[6:] + qb[6:]).max()) <= 1e-5, (name, t)
    finally:
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)


def test_flip_test_on_the_fused_path_matches_reference_process(emu_lib, tmp_path, monkeypatch):
    """`--flip_test` (detector.py:396-399, 469-476, 536-540, `_flip_output` :496-528) on the fused path -- the frame and
    its mirror image as the two frames of one plan, `_sigmoid_output` + `_flip_output` as device operations, decode on the device -- against
    the reference's OWN Detector.process on the same two-frame batch (its DLASeg with the oracle's DCN, its generic_decode): ordered indices /
    classes identical, scores and boxes to 1e-4; MOT heads (hm / wh averaged, the rest from the un-flipped frame) and the nuScenes heads (dep,
    dim averaged, amodel_offset averaged with negated x).  FeatureMaps come back for both frames, as the reference returns them."""
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    ref_shims.install_detector_stubs()
    monkeypatch.setenv("DEFT_HIP_LIB", emu_lib.path)
    from deft_amd import hiplib
    monkeypatch.setattr(hiplib, "_lib", emu_lib)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from dataset.dataset_factory import dataset_factory
        spec = importlib.util.spec_from_file_location("detector", os.path.join(ROOT, "detector.py"))
        shim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shim)
    finally:
        sys.argv = argv
    RD = sys.modules["deft_reference_detector"]
    from deft_amd import checkpoint, detector as FD
    H, W = 64, 96
    torch.set_grad_enabled(False)
    try:
        for dataset, extra in (("mot", ["--ltrb_amodal"]), ("nuscenes", [])):
            sd = dict(O.synth_state_dict(dataset))
            ck = str(tmp_path / ("model_%s.pth" % dataset))
            torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}}, ck)
            opt = opts().parse(["tracking" if dataset == "mot" else "tracking,ddd", "--dataset", dataset, "--gpus", "-1", "--load_model", ck, "--K", "12",
                                "--input_h", str(H), "--input_w", str(W), "--flip_test"] + extra)
            opt = opts().update_dataset_info_and_set_heads(opt, dataset_factory[opt.test_dataset])
            assert opt.flip_test
            x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(21))
            x2 = torch.cat([x, x.flip(3)], 0)                               # Detector.pre_process, detector.py:396-399
            ref = RD.Detector(opt)
            try:
                ro, rd, rmaps = ref.process(x2)
            finally:
                pass
            fd = FD.Detector(opt, checkpoint.load_model_state(ck, opt, log=lambda *_: None))
            fo, fdets, fmaps = fd.process(x2)
            fo1, fdets1, _ = fd.process(x)                                  # one frame in: mirrored on the device
            for got in (fdets, fdets1):
                assert set(got) >= set(rd), (sorted(got), sorted(rd))
                hw = (H // 4) * (W // 4)
                key = lambda d: (np.asarray(d["clses"]).astype(np.int64) * hw + np.round(np.asarray(d["ys"]) * (W // 4) + np.asarray(d["xs"])).astype(np.int64)).tolist()
                # (a map with fewer than K NMS peaks: the reference's top-K fills up with zero-score entries in torch.topk's unspecified tie order)
                npk = int((np.asarray(rd["scores"])[0] > 0).sum())
                assert npk >= 6 and (np.asarray(got["scores"])[0, npk:] <= 0).all()
                assert [r[:npk] for r in key(got)] == [r[:npk] for r in key(rd)], dataset
                for k, v in rd.items():
                    assert np.allclose(np.asarray(got[k], np.float64)[:, :npk], np.asarray(v, np.float64)[:, :npk], atol=1e-4, rtol=1e-5), (dataset, k)
            assert float((fo["hm"].cpu() - ro["hm"]).abs().max()) <= 1e-5
            assert len(fmaps) == 13 and fmaps[0].shape[0] == 2 and tuple(fmaps[0].shape) == tuple(rmaps[0].shape)
            a = fmaps[3][0].unsqueeze(0).to_nchw().cpu()
            assert float((a - rmaps[3][0:1]).abs().max()) <= 1e-4 * max(1.0, float(rmaps[3].abs().max()))
    finally:
        torch.set_grad_enabled(True)
