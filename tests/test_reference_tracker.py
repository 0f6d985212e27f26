"""-m "not gpu" (build container only: needs /root/reference).  DROP-IN check at the level
`src/test.py` consumes: the reference's OWN `Tracker` (utils/tracker.py, FeatureRecorder, STrack,
matching) is run twice over the same short synthetic stream --
  side A: with the reference model (DLASeg + AFE_module on PyTorch-CPU, oracle DCN),
  side B: with deft_amd.integrate.DeftModel (HIP kernels; here through the SIMT emulator) --
and must produce the same tracks: ids, boxes, and the recorder's similarity matrices.  This
exercises seams 2, 3 and 6 of SURVEY.md §8(b) under the reference's real call pattern
(tracker.py:826 forward_feature_extracter, :87 forward_stacker_features per stored frame)."""
import os
import sys

import numpy as np
import pytest
import torch

HAVE_REF = os.path.isdir("/root/reference/src/lib")
pytestmark = [pytest.mark.skipif(not HAVE_REF, reason="/root/reference only exists in the build container"), pytest.mark.slow]


def _results(dets, thr):
    """generic_post_process for an identity affine (post_process.py:29-60): map coords -> input px."""
    out = []
    for i in range(dets["scores"].shape[1]):
        sc = float(dets["scores"][0, i])
        if sc < thr:
            break
        out.append({"score": sc, "class": int(dets["clses"][0, i]) + 1, "bbox": dets["bboxes"][0, i].numpy().astype(np.float32) * 4.0})
    return out


def test_reference_tracker_runs_on_deft_model(emu_lib):
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    from deft_amd import integrate
    ref_shims.install()
    argv, sys.argv = sys.argv, ["test.py", "tracking"]           # utils/tracker.py:139 parses argv at import
    try:
        model_ref, _ = ref_import.build_reference_model("mot", MG.OracleDCN)
        from model.decode import generic_decode
        from opts import opts
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
    finally:
        sys.argv = argv
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
    torch.set_grad_enabled(False)
    try:
        sd = O.synth_state_dict("mot")
        model_ref.load_state_dict(sd, strict=True)
        model_ref.eval()
        model_b = integrate.DeftModel(sd, "mot", K=20, max_object=opt.max_object, device="cpu", lib=emu_lib)
        H, W, T = 32, 64, 3
        frames = [torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(40 + t)) for t in range(T)]

        def run(model):
            BaseTrack._count = 0
            trk = RT.Tracker(opt, model, h=H, w=W)
            per_frame = []
            for x in frames:
                out, fmaps = model(x, None, None)
                out = dict(out[-1])
                out["hm"] = out["hm"].sigmoid()
                dets = generic_decode(out, K=20, opt=opt)
                dets = {k: v.detach().cpu() for k, v in dets.items()}
                res = _results(dets, thr=float(np.sort(dets["scores"][0].numpy())[::-1][5]))      # top-6 detections
                targets = trk.update(res, fmaps)
                per_frame.append(sorted((t.track_id, [float(v) for v in t.tlwh]) for t in targets))
            return per_frame, trk.recorder.all_similarity

        ref_tracks, ref_sim = run(model_ref)
        got_tracks, got_sim = run(model_b)
        assert [[tid for tid, _ in f] for f in got_tracks] == [[tid for tid, _ in f] for f in ref_tracks]
        assert sum(len(f) for f in ref_tracks) > 0, "the synthetic stream must produce tracks"
        for fa, fb in zip(ref_tracks, got_tracks):
            for (_, a), (_, b) in zip(fa, fb):
                assert np.abs(np.array(a) - np.array(b)).max() <= 1e-3
        assert sorted(ref_sim) == sorted(got_sim)
        for f in ref_sim:
            assert sorted(ref_sim[f]) == sorted(got_sim[f])
            for p in ref_sim[f]:
                assert np.abs(np.asarray(ref_sim[f][p]) - np.asarray(got_sim[f][p])).max() <= 1e-4
    finally:
        torch.set_grad_enabled(True)
        for m in ("dcn_v2",):
            sys.modules.pop(m, None)


def test_feature_recorder_mirror_matches_reference(emu_lib):
    """deft_amd.tracker.FeatureRecorder (one launch chain per frame) against the reference's
    FeatureRecorder driven by the reference AFE_module on CPU: same keys, same matrices, incl.
    the decay2 branch (gap >= m_frame) and the eviction of the oldest stored frame."""
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    from deft_amd import integrate, tracker as DT
    ref_shims.install()
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        model_ref, _ = ref_import.build_reference_model("mot", MG.OracleDCN)
        from utils import tracker as RT
    finally:
        sys.argv = argv
    torch.set_grad_enabled(False)
    try:
        sd = O.synth_state_dict("mot")
        model_ref.load_state_dict(sd, strict=True)
        model_ref.eval()

        class M:        # only .AFE is used by the recorder
            AFE = integrate.AfeSeam(sd, 100, "cpu", emu_lib)
        g = torch.Generator().manual_seed(21)
        ref = RT.FeatureRecorder("mot"); ref.max_record_frame = 4
        got = DT.FeatureRecorder("mot", max_record_frame=4)
        for frame, n in [(1, 3), (2, 5), (3, 1), (4, 4), (16, 2), (17, 6)]:      # jump 4 -> 16: decay2 branch; 6 frames > 4 kept
            feats = torch.rand(1, n, 416, generator=g) * 3
            boxes = np.asarray(torch.rand(n, 4, generator=g) * 100)
            ref.update(model_ref, frame, feats, boxes)
            got.update(M, frame, feats, boxes)
            assert list(ref.all_frame_index) == list(got.all_frame_index)
            assert sorted(ref.all_similarity) == sorted(got.all_similarity)
            for f in ref.all_similarity:
                assert sorted(ref.all_similarity[f]) == sorted(got.all_similarity[f])
                for p in ref.all_similarity[f]:
                    a, b = np.asarray(ref.all_similarity[f][p]), np.asarray(got.all_similarity[f][p])
                    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5, (f, p)
        assert got.get_box(17, 2) is not None and got.get_box(1, 0) is None and got.get_features(99) is None
    finally:
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)


def _moving_boxes(t, n=5):
    """n well separated boxes drifting a few px per frame (tlbr + score), input-pixel units."""
    out = []
    for i in range(n):
        x0 = 6.0 + 11.0 * i + 0.8 * t * (1 if i % 2 else -0.5)
        y0 = 4.0 + 2.0 * i + 0.5 * t
        w, h = 7.0 + 0.3 * i + 0.1 * t, 12.0 + 0.5 * i
        out.append({"score": 0.9 - 0.05 * i, "class": 1, "bbox": np.array([x0, y0, x0 + w, y0 + h], np.float32)})
    return out


def test_reference_tracker_with_lstm_seam(emu_lib):
    """Seam 4 under the reference's real call pattern: the reference Tracker with `opt.lstm` on (the 2-D
    builder, tracker.py:408-480) runs once with its own KalmanFilterLSTM and once with
    deft_amd.integrate.KalmanFilterLSTM bound to the same name (tracker.py:144, 301, 661 construct it as
    `KalmanFilterLSTM(opt)`; matching.fuse_motion calls `.gating_distance`).  Same detections and
    FeatureMaps on both sides; tracks, LSTM states and future predictions must agree."""
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    from deft_amd import integrate
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
    finally:
        sys.argv = argv
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
    opt.lstm = True                                            # opts.py:478 turns it off for 2-D datasets; the code path exists
    torch.set_grad_enabled(False)
    ref_cls = RT.KalmanFilterLSTM
    try:
        sd = O.synth_state_dict("mot")
        lsd = O.synth_lstm_state_dict("mot")

        class M:
            AFE = integrate.AfeSeam(sd, opt.max_object, "cpu", emu_lib)
        H, W, T = 32, 64, 6
        chans = [16, 32, 64, 128, 256, 512, 64, 128, 256, 512, 64, 64, 64]
        strides = [1, 2, 4, 8, 16, 32, 4, 8, 16, 32, 4, 4, 4]
        g = torch.Generator().manual_seed(77)
        fmaps = [torch.randn(1, c, H // s, W // s, generator=g) for c, s in zip(chans, strides)]

        def make_ref(o):
            k = ref_cls(o)
            k.model.load_state_dict(lsd, strict=True)
            k.model.eval()
            return k

        def run(factory):
            RT.KalmanFilterLSTM = factory
            RT.STrack.shared_kalman_lstm = factory(opt)
            BaseTrack._count = 0
            trk = RT.Tracker(opt, M, h=H, w=W)
            assert trk.use_lstm
            log = []
            for t in range(T):
                dets = _moving_boxes(t)
                if t == 3:
                    dets = dets[:3]                            # two tracks go unmatched for a frame, then come back
                targets = trk.update(dets, fmaps)
                log.append(sorted((s.track_id, s.tracklet_len, s.tlwh.tolist(), s.hn.reshape(-1).tolist(),
                                   {k: v.tolist() for k, v in s.future_predictions.items()}) for s in targets))
            return log

        ref = run(make_ref)
        got = run(lambda o: integrate.KalmanFilterLSTM(o, lsd, device="cpu", lib=emu_lib))
        assert max(s[1] for s in ref[-1]) >= 2, "tracks must have been updated (non-first-time feature path)"

        # the per-frame forms: recorder mirror + device-side Tracker.get_similarity + ONE motion launch per frame
        from deft_amd import tracker as DT
        bank = DT.MotionBank(integrate.KalmanFilterLSTM(opt, lsd, device="cpu", lib=emu_lib))
        undo = DT.install_batched_motion(RT.STrack, bank)
        ref_get, ref_rec = RT.Tracker.get_similarity, RT.FeatureRecorder
        RT.Tracker.get_similarity = DT.get_similarity
        RT.FeatureRecorder = DT.FeatureRecorder                   # Tracker.__init__ (tracker.py:651) builds it by this name
        from deft_amd import association
        unbind = association.bind(RT.matching)                    # vectorised fuse_motion / linear_assignment / IoU
        try:
            batched = run(lambda o: integrate.KalmanFilterLSTM(o, lsd, device="cpu", lib=emu_lib))
        finally:
            undo()
            unbind()
            RT.Tracker.get_similarity, RT.FeatureRecorder = ref_get, ref_rec
        assert bank.launches <= T, "one motion launch per frame, not one per track"
        assert "future_predictions" not in RT.STrack.__dict__

        for fa, fb in list(zip(ref, got)) + list(zip(ref, batched)):
            assert [(a[0], a[1]) for a in fa] == [(b[0], b[1]) for b in fb]
            for a, b in zip(fa, fb):
                assert np.abs(np.array(a[2]) - np.array(b[2])).max() <= 1e-4
                if len(b[3]) == len(a[3]) and any(b[3]):          # the batched form keeps (h, c) in the bank, not on the track
                    assert np.abs(np.array(a[3]) - np.array(b[3])).max() <= 1e-5
                assert sorted(a[4]) == sorted(b[4])
                for k in a[4]:
                    assert np.abs(np.array(a[4][k]) - np.array(b[4][k])).max() <= 1e-3 * max(1.0, np.abs(a[4][k]).max())
    finally:
        RT.KalmanFilterLSTM = ref_cls
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)


def test_reference_tracker_with_vectorised_association(emu_lib):
    """The reference's default 2-D configuration (opts.py:478: Kalman filter, no LSTM): its Tracker as written
    against the same Tracker with the per-frame forms bound in -- recorder mirror (one affinity chain per frame),
    device-side get_similarity, vectorised fuse_motion / linear_assignment / IoU (deft_amd.association)."""
    import deft_oracle as O
    import make_golden as MG
    import ref_import
    import ref_shims
    from deft_amd import association, integrate, tracker as DT
    ref_shims.install()
    ref_import.install_stubs(MG.OracleDCN)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
    finally:
        sys.argv = argv
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
    assert not opt.lstm
    torch.set_grad_enabled(False)
    try:
        sd = O.synth_state_dict("mot")

        class M:
            AFE = integrate.AfeSeam(sd, opt.max_object, "cpu", emu_lib)
        H, W, T = 32, 64, 7
        chans = [16, 32, 64, 128, 256, 512, 64, 128, 256, 512, 64, 64, 64]
        strides = [1, 2, 4, 8, 16, 32, 4, 8, 16, 32, 4, 4, 4]
        g = torch.Generator().manual_seed(78)
        fmaps = [torch.randn(1, c, H // s, W // s, generator=g) for c, s in zip(chans, strides)]

        def run():
            BaseTrack._count = 0
            trk = RT.Tracker(opt, M, h=H, w=W)
            log = []
            for t in range(T):
                dets = _moving_boxes(t)
                if t in (3, 4):
                    dets = dets[1:4]
                targets = trk.update(dets, fmaps)
                log.append(sorted((s.track_id, s.tracklet_len, s.tlwh.tolist(), s.mean.tolist()) for s in targets))
            return log

        ref = run()
        ref_get, ref_rec = RT.Tracker.get_similarity, RT.FeatureRecorder
        RT.Tracker.get_similarity = DT.get_similarity
        RT.FeatureRecorder = DT.FeatureRecorder
        unbind = association.bind(RT.matching)
        M.AFE.host_copy = False                                   # affinity blocks stay on the device
        try:
            got = run()
        finally:
            unbind()
            M.AFE.host_copy = True
            RT.Tracker.get_similarity, RT.FeatureRecorder = ref_get, ref_rec
        assert max(s[1] for s in ref[-1]) >= 2
        for fa, fb in zip(ref, got):
            assert [(a[0], a[1]) for a in fa] == [(b[0], b[1]) for b in fb]
            for a, b in zip(fa, fb):
                assert np.abs(np.array(a[2]) - np.array(b[2])).max() <= 1e-6
                assert np.abs(np.array(a[3]) - np.array(b[3])).max() <= 1e-6
    finally:
        torch.set_grad_enabled(True)
        sys.modules.pop("dcn_v2", None)
