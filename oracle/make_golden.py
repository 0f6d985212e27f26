"""TEST INFRASTRUCTURE ONLY.  Pins oracle/deft_oracle.py against the reference's
own modules and writes the golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference):

    python oracle/make_golden.py

What is pinned here (reference modules imported unchanged via ref_import.py):
  * DLASeg forward (dla.py:758-817, base_model.py:111-132) incl. all 13 FeatureMaps
    -- with the reference's DeformConv/IDAUp/DLAUp code driving `dcn_v2.DCN`
       bound to the oracle's DCN restatement (DCNv2 itself stays UNPINNED: its
       source is not in /root/reference).
  * AFE_module.forward_feature_extracter / forward_stacker_features (AFE.py:88-160)
  * model.decode.generic_decode (decode.py:102) on sigmoid'ed heads
  * KalmanFilterLSTM.predict (kalman_filter_lstm.py:65-78)
  * utils.image.convert_detection (image.py:391-412)
The script asserts oracle == reference (max-abs <= 2e-5 on floats, exact on
indices) and stores the REFERENCE outputs as the fixtures.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import deft_oracle as O  # noqa: E402
import ref_import  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")


class OracleDCN(torch.nn.Module):
    """Stands in for the un-vendored `dcn_v2.DCN` (ctor per dla.py:652-660)."""

    def __init__(self, cin, cout, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        assert tuple(kernel_size) == (3, 3) and stride == 1 and padding == 1 and dilation == 1
        self.weight = torch.nn.Parameter(torch.zeros(cout, cin, 3, 3))
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.conv_offset_mask = torch.nn.Conv2d(cin, 27, 3, 1, 1)

    def forward(self, x):
        return O.dcn_v2_forward(x, self.conv_offset_mask.weight, self.conv_offset_mask.bias,
                                self.weight, self.bias)


def maxabs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def sample_map(t, n=64, seed=0):
    """Deterministic sparse sample of a tensor (keeps fixtures small)."""
    flat = t.reshape(-1)
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, flat.numel(), (n,), generator=g)
    return idx.numpy(), flat[idx].numpy()


def run(dataset, H, W, tag):
    torch.set_grad_enabled(False)
    model, opt = ref_import.build_reference_model(dataset, OracleDCN)
    sd = O.synth_state_dict(dataset)
    missing = model.load_state_dict(sd, strict=True)
    print(tag, "state_dict loaded strict:", missing)
    model.eval()
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0))
    # reference .cuda() calls are gated on torch.cuda.is_available() (False here)
    ref_out, ref_maps = model(x)
    ref_out = ref_out[-1]
    ora_out, ora_maps = O.dlaseg_forward(x, sd, dataset)
    fix = {"H": H, "W": W}
    for k in range(13):
        d = maxabs(ref_maps[k], ora_maps[k])
        print("  fmap %2d %s maxabs %.3e  (|ref| max %.3f)" % (k, tuple(ref_maps[k].shape), d, ref_maps[k].abs().max()))
        assert d <= 2e-5 * max(1.0, float(ref_maps[k].abs().max()))
        idx, val = sample_map(ref_maps[k], 96, seed=k)
        fix["fmap%d_idx" % k] = idx; fix["fmap%d_val" % k] = val
        fix["fmap%d_sum" % k] = np.float64(ref_maps[k].double().sum())
        fix["fmap%d_abs" % k] = np.float64(ref_maps[k].double().abs().sum())
    for h in ref_out:
        d = maxabs(ref_out[h], ora_out[h])
        print("  head %s maxabs %.3e" % (h, d))
        assert d <= 2e-5 * max(1.0, float(ref_out[h].abs().max()))
        idx, val = sample_map(ref_out[h], 128, seed=100)
        fix["head_%s_idx" % h] = idx; fix["head_%s_val" % h] = val
    # ---- decode (detector.py:486-494 + decode.py:102) ----
    from model.decode import generic_decode
    sg = ref_out["hm"].sigmoid()
    npk = int((torch.nn.functional.max_pool2d(sg, 3, 1, 1) == sg).sum())
    K = 100 if npk >= 150 else max(4, min(20, npk // 2))
    print("  peaks %d -> K %d" % (npk, K))
    ref_sig = {k: v.clone() for k, v in ref_out.items()}
    ref_sig["hm"] = ref_sig["hm"].sigmoid_()
    if "dep" in ref_sig:
        ref_sig["dep"] = 1.0 / (ref_sig["dep"].sigmoid() + 1e-6) - 1.0
    ref_dets = generic_decode(ref_sig, K=K, opt=opt)
    ora_dets = O.generic_decode(O.sigmoid_output(ora_out), K=K)
    # the oracle's own path must reproduce the reference's indices bit-exactly
    hw = (H // 4) * (W // 4)
    ref_inds = (ref_dets["ys"] * (W // 4) + ref_dets["xs"]).long()
    assert torch.equal(ref_inds, ora_dets["inds"]), "top-k indices differ (oracle vs reference)"
    for k in ["scores", "clses", "xs", "ys", "bboxes", "tracking"]:
        d = maxabs(ref_dets[k], ora_dets[k])
        print("  dets %s maxabs %.3e" % (k, d))
        assert d <= 1e-4
        fix["det_" + k] = ref_dets[k].numpy()
    fix["det_inds"] = ref_inds.numpy(); fix["det_K"] = K
    gap = (ref_dets["scores"][0, :-1] - ref_dets["scores"][0, 1:]).min()
    print("  min adjacent top-k score gap %.3e" % float(gap))
    # ---- embeddings (AFE.py:88) ----
    N = 12
    g = torch.Generator().manual_seed(2)
    centers = (torch.rand(1, N, 1, 1, 2, generator=g) * 2 - 1)
    centers[0, 0] = torch.tensor([-1.0, -1.0]); centers[0, 1] = torch.tensor([1.0, 1.0])  # border cases
    centers[0, 2] = torch.tensor([0.9999, -0.9999])
    ref_emb = model.AFE.forward_feature_extracter(ref_maps, centers)
    ora_emb = O.afe_extract(ora_maps, centers, sd)
    d = maxabs(ref_emb, ora_emb)
    print("  embed %s maxabs %.3e" % (tuple(ref_emb.shape), d))
    assert d <= 2e-5 * max(1.0, float(ref_emb.abs().max()))
    fix["emb_centers"] = centers.numpy(); fix["emb"] = ref_emb.numpy()
    # ---- affinity (AFE.py:110) ----
    D = ref_emb.shape[2]
    g = torch.Generator().manual_seed(5)
    for n, (P, Q) in enumerate([(5, 7), (12, 12), (1, 3), (9, 2)]):
        xp = torch.randn(1, P, D, generator=g).abs() * 3.0
        xn = torch.randn(1, Q, D, generator=g).abs() * 3.0
        ref_a = model.AFE.forward_stacker_features(xp, xn, False)
        ora_a = O.afe_affinity(xp, xn, sd, opt.max_object)
        d = maxabs(ref_a, ora_a)
        print("  affinity %dx%d maxabs %.3e range [%.4f, %.4f]" % (P, Q, d, ref_a.min(), ref_a.max()))
        assert d <= 1e-6
        fix["aff%d_xp" % n] = xp.numpy(); fix["aff%d_xn" % n] = xn.numpy(); fix["aff%d" % n] = ref_a
    # embeddings-driven affinity (realistic magnitudes)
    ref_a = model.AFE.forward_stacker_features(ref_emb[:, :7], ref_emb[:, 5:], False)
    ora_a = O.afe_affinity(ora_emb[:, :7], ora_emb[:, 5:], sd, opt.max_object)
    assert maxabs(ref_a, ora_a) <= 1e-5
    fix["aff_emb"] = ref_a
    np.savez_compressed(os.path.join(GOLD, "forward_%s.npz" % tag), **fix)
    return model


def run_lstm(dataset):
    from utils.tracking_utils.kalman_filter_lstm import KalmanFilterLSTM
    opt = ref_import.make_opt(dataset)
    kf = KalmanFilterLSTM(opt)
    lsd = O.synth_lstm_state_dict(dataset)
    kf.model.load_state_dict(lsd, strict=True)
    kf.model.eval()
    nin = lsd["lstm.weight_ih_l0"].shape[1]
    T, steps = 6, 4
    g = torch.Generator().manual_seed(7)
    h = torch.zeros(T, 128); c = torch.zeros(T, 128)
    xs = torch.randn(steps, T, nin, generator=g)
    fix = {"xs": xs.numpy()}
    with torch.no_grad():
        hr = h.clone(); cr = c.clone()
        for s in range(steps):
            ho, co, po = O.lstm_predict(h, c, xs[s], lsd)
            preds = []
            for t in range(T):
                hn, cn, pr = kf.predict(hr[t].view(1, 1, 128), cr[t].view(1, 1, 128), xs[s, t].view(1, 1, nin))
                hr[t] = hn.view(128); cr[t] = cn.view(128)
                preds.append(np.stack([pr[i + 1] for i in range(kf.MAX_dis_fut)]))
            preds = np.stack(preds)
            d = max(maxabs(hr, ho), maxabs(cr, co), maxabs(preds, po))
            print("  lstm[%s] step %d maxabs %.3e" % (dataset, s, d))
            assert d <= 2e-6
            h, c = ho, co
            fix["h%d" % s] = hr.numpy().copy(); fix["c%d" % s] = cr.numpy().copy(); fix["p%d" % s] = preds
    np.savez_compressed(os.path.join(GOLD, "lstm_%s.npz" % dataset), **fix)


def run_convert_detection():
    from utils.image import convert_detection
    g = np.random.RandomState(3)
    boxes = g.rand(9, 4) * 400
    boxes[:, 2:] += boxes[:, :2]
    ref = convert_detection(boxes.copy(), 608.0, 1088.0)
    ora = O.convert_detection(boxes.copy(), 608.0, 1088.0)
    assert torch.equal(ref, ora)
    np.savez_compressed(os.path.join(GOLD, "convert_detection.npz"), boxes=boxes, centers=ref.numpy())


def _tracker_module():
    """utils/tracker.py parses sys.argv at import (tracker.py:139) and needs lap/cython_bbox/numba stand-ins."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(OracleDCN)
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from utils import tracker as RT
    finally:
        sys.argv = argv
    return RT, opts


def run_motion(dataset):
    """STrack.update_lstm_features / _ddd (tracker.py:408-580) of the reference, driven directly on one track:
    features (captured at the predict call), LSTM state and future_predictions per step."""
    RT, opts = _tracker_module()
    opt = opts().parse(["tracking" if dataset != "nuscenes" else "tracking,ddd", "--dataset", dataset, "--gpus", "-1"])
    opt.lstm = True
    lsd = O.synth_lstm_state_dict(dataset)
    kf = RT.KalmanFilterLSTM(opt)
    kf.model.load_state_dict(lsd, strict=True)
    kf.model.eval()
    seen = []
    real_predict = kf.predict

    def spy(h0, c0, new_features):
        seen.append(new_features.reshape(-1).numpy().copy())
        return real_predict(h0, c0, new_features)
    kf.predict = spy
    RT.STrack.shared_kalman_lstm = kf
    ddd = dataset == "nuscenes"
    g = np.random.RandomState(11 + ddd)
    frames = [1, 2, 3, 5, 6, 9, 10]                      # gaps: the velocities divide by the frame distance
    if ddd:
        boxes = np.cumsum(g.randn(len(frames), 7) * 0.3, 0) + np.array([1.6, 1.9, 4.5, 3.0, 1.2, 25.0, 0.4])
    else:
        boxes = np.cumsum(g.randn(len(frames), 4) * 3.0, 0) + np.array([300.0, 180.0, 42.0, 110.0])
    trk = RT.STrack(np.array([0.0, 0.0, 1.0, 1.0]), 0.9, RT.Node(1, 0), 30, use_lstm=True, opt=opt,
                    ddd_bbox=boxes[0] if ddd else None, depth=1.0)
    mo = O.MotionTrack(lsd, ddd)
    fix = {"frames": np.array(frames), "boxes": boxes}
    with torch.no_grad():
        for s, (f, b) in enumerate(zip(frames, boxes)):
            trk.frame_id = f
            if ddd:
                trk.update_lstm_features_ddd(b.copy())
            else:
                trk.update_lstm_features(b.copy())
            ora = mo.update(b, f)
            ref_fut = np.stack([np.asarray(trk.future_predictions[k]) for k in sorted(trk.future_predictions)])
            ora_fut = np.stack([np.asarray(ora[k]) for k in sorted(ora)])
            assert np.array_equal(seen[-1], mo.features), "motion features must be bit-identical"
            d = max(maxabs(trk.hn.reshape(-1), mo.h.reshape(-1)), float(np.abs(ref_fut - ora_fut).max() / max(1.0, np.abs(ref_fut).max())))
            print("  motion[%s] step %d frame %d maxabs %.3e" % (dataset, s, f, d))
            assert d <= 2e-6 and ref_fut.dtype == ora_fut.dtype
            fix["feat%d" % s] = seen[-1]; fix["fut%d" % s] = ref_fut
            fix["h%d" % s] = trk.hn.reshape(-1).numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "motion_%s.npz" % dataset), **fix)


def run_track_similarity():
    """Tracker.get_similarity (tracker.py:663-688) of the reference on hand-built recorder contents: tracks with
    0, 1, 2, 4, 5, 6 and 9 nodes (all three branches of STrack.get_similarity), a node older than
    max_track_node, both decay branches."""
    RT, _ = _tracker_module()
    g = np.random.RandomState(5)
    frame, ndet = 60, 7
    prev = [8, 52, 54, 55, 56, 57, 58, 59]
    counts = {8: 3, 52: 4, 54: 6, 55: 5, 56: 6, 57: 4, 58: 6, 59: 5}
    sim = {}
    for p in prev:
        gap = frame - p
        delta = pow(RT.decay, gap / 3.0) if gap < 10 else pow(RT.decay2, gap / 3.0)
        sim[p] = g.rand(counts[p], ndet + 1).astype(np.float32) * delta
    tracks_nodes = [
        [],
        [(59, 1)],
        [(8, 2)],                                           # only node is >= max_track_node frames old -> zero row
        [(57, 0), (59, 4)],
        [(8, 0), (55, 1), (56, 2), (58, 3), (59, 0)],       # old node dropped -> 4 rows
        [(54, 5), (55, 4), (56, 5), (57, 3), (58, 5)],      # mm+1 rows: all used
        [(52, 1), (54, 0), (55, 0), (56, 0), (58, 0), (59, 2)],             # > mm+1: last 4
        [(52, 3), (54, 1), (54, 2), (55, 2), (56, 1), (57, 1), (57, 2), (58, 1), (59, 3)],
    ]

    class Rec:
        all_similarity = {frame: sim}

    class Self:
        recorder = Rec()
    pool = []
    for nodes in tracks_nodes:
        t = RT.STrack.__new__(RT.STrack)
        t.nodes = [RT.Node(f, i) for f, i in nodes]
        t.dataset = "mot"
        pool.append(t)
    fix = {"frame": frame, "ndet": ndet, "prev": np.array(prev)}
    for p in prev:
        fix["sim_%d" % p] = sim[p]
    fix["nodes"] = np.array([(t, f, i) for t, nodes in enumerate(tracks_nodes) for f, i in nodes])
    for ds in ("mot", "nuscenes"):
        for t in pool:
            t.dataset = ds
        ref = RT.Tracker.get_similarity(Self(), frame, pool, ndet)
        ora = O.track_similarity(sim, tracks_nodes, frame, ndet, ds)
        assert ref.dtype == ora.dtype and np.array_equal(ref, ora), "track similarity must be bit-identical"
        fix["out_" + ds] = ref
    for t in pool:
        t.nodes = []                                         # STrack.__del__ walks .nodes
    np.savez_compressed(os.path.join(GOLD, "track_similarity.npz"), **fix)
    print("  track similarity: %d tracks, oracle == reference" % len(pool))


def run_association():
    """matching.fuse_motion / fuse_motion_ddd (matching.py:311-415) of the reference with the reference's own
    KalmanFilter / KalmanFilterLSTM.gating_distance, on mock tracks: inputs and outputs as fixtures for
    deft_amd.association (vectorised).  linear_assignment/ious go through third-party packages that are absent
    here (lap, cython_bbox): unpinned, checked against brute force in tests/test_association.py."""
    RT, opts = _tracker_module()
    from types import SimpleNamespace
    from utils import matching
    from utils.tracking_utils.kalman_filter import KalmanFilter
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1"])
    g = np.random.RandomState(17)
    T, N = 7, 9
    det_tlwh = np.abs(g.randn(N, 4)) * np.array([300, 150, 20, 40]) + np.array([50, 30, 15, 30])
    dets = [SimpleNamespace(to_xyah=(lambda b=b: RT.STrack.tlwh_to_xyah(b))) for b in det_tlwh]
    fix = {"det_tlwh": det_tlwh}

    def spd(n):
        a = g.randn(n, n)
        return a @ a.T + n * np.eye(n)
    # Kalman branch (use_lstm=False): mean [8], covariance [8,8]; means placed near some detections
    means = np.stack([np.r_[RT.STrack.tlwh_to_xyah(det_tlwh[t % N]) + g.randn(4) * np.array([6, 6, 0.01, 2]) * (1 + 4 * (t % 3 == 0)), g.randn(4)] for t in range(T)])
    covs = np.stack([spd(8) * (4.0 if t % 2 else 40.0) for t in range(T)])
    tracks = [SimpleNamespace(mean=means[t], covariance=covs[t]) for t in range(T)]
    cost = g.rand(T, N)
    fix.update(kal_mean=means, kal_cov=covs, kal_cost=cost,
               kal_out=matching.fuse_motion(KalmanFilter(), cost.copy(), tracks, dets, frame_id=5, use_lstm=False))
    # LSTM branch: tracks with >= 300 observations (maha on the prediction + np.cov covariance) and younger ones
    kfl = RT.KalmanFilterLSTM(opt)
    preds = np.stack([RT.STrack.tlwh_to_xyah(det_tlwh[(t + 2) % N]) + g.randn(4) * np.array([5, 5, 0.01, 2]) for t in range(T)]).astype(np.float32)
    nobs = np.array([300, 5, 450, 299, 1, 300, 20])
    cov4 = np.stack([spd(4) * 3.0 for _ in range(T)])
    tracks = [SimpleNamespace(observations=[0] * int(nobs[t]), covariance=cov4[t], prediction_at_frame=(lambda f, t=t: preds[t])) for t in range(T)]
    cost = g.rand(T, N)
    fix.update(lstm_pred=preds, lstm_nobs=nobs, lstm_cov=cov4, lstm_cost=cost,
               lstm_out=matching.fuse_motion(kfl, cost.copy(), tracks, dets, frame_id=5, use_lstm=True))
    assert np.isinf(fix["kal_out"]).any() and np.isinf(fix["lstm_out"]).any() and np.isfinite(fix["lstm_out"]).any()
    # 3-D branch
    det_ddd = np.abs(g.randn(N, 7)) + np.array([1.5, 1.8, 4.2, 0, 1, 20, 0]) + g.randn(N, 7) * np.array([0, 0, 0, 8, 0.3, 10, 1])
    trk_ddd = det_ddd[g.permutation(N)[:T]] + g.randn(T, 7) * np.array([0.1, 0.1, 0.1, 3, 0.2, 3, 0.1])
    depth = trk_ddd[:, 5].copy(); depth[0] = 80.0
    dets3 = [SimpleNamespace(ddd_bbox=b) for b in det_ddd]
    tracks3 = [SimpleNamespace(ddd_bbox=trk_ddd[t], depth=depth[t], covariance=np.eye(7)) for t in range(T)]
    cost = g.rand(T, N)
    fix.update(ddd_det=det_ddd, ddd_trk=trk_ddd, ddd_depth=depth, ddd_cost=cost)
    for cls in ("pedestrian", "car"):
        fix["ddd_out_" + cls] = matching.fuse_motion_ddd(kfl, cost.copy(), tracks3, dets3, frame_id=5, classe_name=cls)
        assert np.isinf(fix["ddd_out_" + cls]).any()
    # opt.lstm off: Tracker.kalman_filter is the plain KalmanFilter (tracker.py:652), whose "gaussian" distance is another formula
    # (kalman_filter.py:271-273: squared, all seven components); tracks close enough for its 5 / 10 gates
    trk_near = det_ddd[g.permutation(N)[:T]] + g.randn(T, 7) * np.array([0.3, 0.3, 0.3, 1.2, 0.4, 1.2, 0.2])
    tracks3n = [SimpleNamespace(ddd_bbox=trk_near[t], depth=depth[t], covariance=np.eye(7)) for t in range(T)]
    fix["ddd_trk_near"] = trk_near
    for cls in ("pedestrian", "car"):
        fix["ddd_out_kf_" + cls] = matching.fuse_motion_ddd(RT.KalmanFilter(), cost.copy(), tracks3n, dets3, frame_id=5, classe_name=cls)
        assert np.isinf(fix["ddd_out_kf_" + cls]).any() and np.isfinite(fix["ddd_out_kf_" + cls]).any()
    np.savez_compressed(os.path.join(GOLD, "association.npz"), **fix)
    print("  association fixtures written (reference fuse_motion / fuse_motion_ddd)")


def run_iou_ddd():
    """matching.iou_ddd_distance (matching.py:107-131: convert_3dbox_to_8corner, polygon_clip, scipy ConvexHull, iou3d) of the reference
    on random (h, w, l, x, y, z, rot_y) boxes -- near-duplicates, partial overlaps, disjoint pairs, different heights -- as a fixture
    for deft_amd.association.iou_ddd_distance (csrc/assoc.hip)."""
    RT, opts = _tracker_module()
    from types import SimpleNamespace
    from utils import matching
    g = np.random.RandomState(23)
    N, T = 40, 33
    det = np.abs(g.randn(N, 7)) * np.array([0.3, 0.3, 0.8, 0, 0, 0, 0]) + np.array([1.5, 1.8, 4.2, 0, 1, 20, 0]) \
        + g.randn(N, 7) * np.array([0, 0, 0, 6, 0.4, 8, 1.2])
    trk = det[g.randint(0, N, T)] + g.randn(T, 7) * np.array([0.1, 0.1, 0.3, 1.0, 0.3, 1.5, 0.3]) * np.where(g.rand(T, 1) < 0.7, 1.0, 0.02)      # (exactly equal boxes make the reference's clip divide by zero: qhull raises)
    trk[:4] = det[:4] + g.randn(4, 7) * 1e-3                          # near-duplicates: IoU close to 1
    trk[4, 4] += 5.0                                                  # the same ground rectangle, no vertical overlap
    out = matching.iou_ddd_distance([SimpleNamespace(ddd_bbox=b) for b in trk], [SimpleNamespace(ddd_bbox=b) for b in det])
    assert out.dtype == np.float32 and (out < 0.2).any() and (out == 1.0).any() and ((out > 0.3) & (out < 0.9)).any()
    np.savez_compressed(os.path.join(GOLD, "iou_ddd.npz"), det=det, trk=trk, out=out)
    print("  iou_ddd: %d x %d pairs, %d overlapping (reference iou_ddd_distance)" % (T, N, int((out < 1.0).sum())))


def run_result_writers():
    """src/test.py:322-342 `write_results` (the reference's own function, exec'd out of test.py's source: importing the module would run
    its argument parser and needs cv2 / the datasets) on synthetic per-frame track lists, MOT and KITTI formats: the text it writes as a
    fixture for deft_amd.results.write_results."""
    import ast
    import tempfile
    src = open(os.path.join(os.path.dirname(ref_import.REF_LIB), "test.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "write_results"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "test.py:write_results", "exec"), ns)
    g = np.random.RandomState(31)
    results = []
    for frame in range(1, 7):
        n = int(g.randint(0, 5))
        tlwhs = [np.array([g.rand() * 900, g.rand() * 500, 20 + g.rand() * 80, 40 + g.rand() * 160]) for _ in range(n)]
        if frame == 3 and n:
            tlwhs[0] = tlwhs[0].astype(np.float32)                   # a float32 row formats with fewer digits
        ids = [int(v) for v in g.randint(1, 40, n)]
        if frame == 4 and n:
            ids[0] = -1                                              # skipped by the writer
        results.append((frame, tlwhs, ids))
    fix = {"nframes": len(results)}
    for k, (frame, tlwhs, ids) in enumerate(results):
        fix["f%d_frame" % k] = np.array(frame); fix["f%d_ids" % k] = np.array(ids, np.int64)
        fix["f%d_tlwh" % k] = np.array([np.asarray(t, np.float64) for t in tlwhs]).reshape(-1, 4)
        fix["f%d_f32" % k] = np.array([t.dtype == np.float32 for t in tlwhs], bool)
    for data_type in ("mot", "kitti_tracking"):
        with tempfile.NamedTemporaryFile("r", suffix=".txt") as tmp:
            ns["write_results"](tmp.name, [(f, list(t), list(i)) for f, t, i in results], data_type)
            fix["text_" + data_type] = np.array(open(tmp.name).read())
    np.savez_compressed(os.path.join(GOLD, "result_writers.npz"), **fix)
    print("  result writers: %d frames, %d + %d characters (reference write_results)" % (len(results), len(str(fix["text_mot"])), len(str(fix["text_kitti_tracking"]))))


def run_postprocess():
    """utils.post_process.generic_post_process (post_process.py:29-112) and utils.ddd_utils.nms (ddd_utils.py:178-245) of the
    reference on synthetic decoded detections (MOT heads and the nuScenes 3-D heads): inputs and outputs as fixtures for
    deft_amd.postprocess.  cv2.getAffineTransform is the float64 3-point solve of tests/ref_shims.py (cv2 is absent: unpinned)."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    import ref_shims
    from types import SimpleNamespace
    ref_shims.install()
    ref_import.install_stubs(OracleDCN)
    ref_shims.install_detector_stubs()
    from utils.post_process import generic_post_process
    from utils.ddd_utils import nms
    g = np.random.RandomState(5)
    fix = {}
    for tag, ddd, (Hh, Ww, oh, ow) in (("mot", False, (1080, 1920, 152, 272)), ("nusc", True, (900, 1600, 112, 200))):
        K = 40
        scores = np.sort(g.rand(1, K).astype(np.float32) * 0.9)[:, ::-1].copy()
        cts = (g.rand(1, K, 2) * np.array([ow, oh])).astype(np.float32)
        wh = (g.rand(1, K, 2) * 30 + 2).astype(np.float32)
        dets = {"scores": scores, "clses": g.randint(0, 10 if ddd else 1, (1, K)).astype(np.float32), "cts": cts,
                "bboxes": np.concatenate([cts - wh / 2, cts + wh / 2], 2).astype(np.float32), "tracking": g.randn(1, K, 2).astype(np.float32) * 3}
        calib = np.array([[1266.4, 0.0, 816.3, 0.0], [0.0, 1266.4, 491.5, 0.0], [0.0, 0.0, 1.0, 0.0]], np.float32)
        if ddd:
            dets.update(dep=(g.rand(1, K, 1) * 60 + 3).astype(np.float32), dim=(g.rand(1, K, 3) * 3 + 0.5).astype(np.float32),
                        rot=g.randn(1, K, 8).astype(np.float32), amodel_offset=g.randn(1, K, 2).astype(np.float32))
        c = np.array([Ww / 2.0, Hh / 2.0], np.float32); s = np.float32(max(Hh, Ww))
        opt = SimpleNamespace(out_thresh=0.25)
        ref = generic_post_process(opt, {k: v.copy() for k, v in dets.items()}, [c], [s], oh, ow, 10, [calib], Hh, Ww)[0]
        for k, v in dets.items():
            fix["%s_in_%s" % (tag, k)] = v
        fix["%s_c" % tag], fix["%s_s" % tag], fix["%s_hw" % tag], fix["%s_calib" % tag] = c, s, np.array([oh, ow]), calib
        fix["%s_n" % tag] = np.array(len(ref))
        for key in ("score", "class", "ct", "bbox", "tracking") + (("dep", "dim", "alpha", "loc", "rot_y") if ddd else ()):
            fix["%s_out_%s" % (tag, key)] = np.array([np.asarray(r[key]).reshape(-1) for r in ref])
        assert 5 < len(ref) < K
    for case, (n, ov) in enumerate(((30, 0.8), (17, 0.7), (1, 0.8), (60, 0.5))):
        ctr = g.rand(n, 2) * 200
        sz = g.rand(n, 2) * 60 + 10
        boxes = np.concatenate([ctr - sz / 2, ctr + sz / 2], 1)
        boxes[n // 2:] = boxes[:n - n // 2] + g.randn(n - n // 2, 4) * 2          # near-duplicates to suppress
        sc = g.rand(n)
        keep, count = nms(torch.from_numpy(boxes), torch.from_numpy(sc), overlap=ov)
        fix["nms%d_boxes" % case], fix["nms%d_scores" % case], fix["nms%d_overlap" % case] = boxes, sc, np.array(ov)
        fix["nms%d_keep" % case], fix["nms%d_count" % case] = keep.numpy(), np.array(count)
    np.savez_compressed(os.path.join(GOLD, "postprocess.npz"), **fix)
    print("  post-process fixtures written (reference generic_post_process / nms)")


def run_detector_trace(lstm):
    """The reference's OWN `Detector.run` (detector.py:112-344: pre-processed branch of src/test.py:213, its post-processing, its
    Tracker) over 6 synthetic frames with the reference model on CPU, and a TRACE of every call it makes into the seams this
    repository replaces -- what went in and what came out:
        process()                               frame -> decoded detections (K rows)
        model.AFE.forward_feature_extracter     detection centres -> embeddings
        FeatureRecorder.update                  -> the frame's (decayed) similarity blocks against every stored frame
        Tracker.get_similarity                  (tracks' node lists, #detections) -> the tracks x detections matrix
        STrack.update_lstm_features (lstm)      (track, tlwh, frame) -> LSTM features and future boxes
    plus the tracks it returns.  tests replay the trace against the HIP path on cuda:0 in the same order
    (tests/test_gpu_parity.py::test_composed_dropin_replays_reference_trace) -- the GPU box has no /root/reference."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    import importlib
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(OracleDCN)
    ref_shims.install_detector_stubs()
    argv, sys.argv = sys.argv, ["test.py", "tracking"]
    try:
        from opts import opts
        from dataset.dataset_factory import dataset_factory
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
        RD = importlib.import_module("detector")
    finally:
        sys.argv = argv
    tag = "mot_lstm" if lstm else "mot"
    sd = dict(O.synth_state_dict("mot"))
    sd["ltrb_amodal.2.weight"] = sd["ltrb_amodal.2.weight"] * 0.05           # boxes of ~10 x 16 map pixels (random heads give negative extents)
    sd["ltrb_amodal.2.bias"] = torch.tensor([-5.0, -8.0, 5.0, 8.0])
    ck = os.path.join(GOLD, "_trace_ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}}, ck)
    H, W, K, T = 64, 96, 8, 6
    opt = opts().parse(["tracking", "--dataset", "mot", "--gpus", "-1", "--load_model", ck, "--K", str(K), "--ltrb_amodal",
                        "--input_h", str(H), "--input_w", str(W)])
    opt = opts().update_dataset_info_and_set_heads(opt, dataset_factory[opt.test_dataset])
    opt.out_thresh = 0.0
    opt.lstm = bool(lstm)
    fix = {"H": H, "W": W, "K": K, "T": T, "lstm": int(lstm), "seeds": np.arange(10, 10 + T)}
    real_sync, torch.cuda.synchronize = torch.cuda.synchronize, (lambda *a, **k: None)          # detector.py:188, 534 call it unconditionally
    lsd = O.synth_lstm_state_dict("mot")
    saved = (RD.Detector.process, RT.Tracker.get_similarity, RT.STrack.update_lstm_features, RT.KalmanFilterLSTM)
    cur = {"t": 0, "gs": 0, "mo": 0}
    try:
        if lstm:
            class KF(saved[3]):
                def __init__(self, o):
                    super().__init__(o)
                    self.model.load_state_dict(lsd, strict=True); self.model.eval()
            RT.KalmanFilterLSTM = KF
            RT.STrack.shared_kalman_lstm = KF(opt)

        def process(self, images, *a, **k):
            r = saved[0](self, images, *a, **k)
            for key, v in r[1].items():
                fix["t%d_det_%s" % (cur["t"], key)] = np.asarray(v.detach().cpu() if torch.is_tensor(v) else v)
            return r

        def get_similarity(self, frame_index, strack_pool, num_detections):
            out = saved[1](self, frame_index, strack_pool, num_detections)
            k = "t%d_gs%d" % (cur["t"], cur["gs"]); cur["gs"] += 1
            fix[k + "_out"] = np.asarray(out, np.float64)
            fix[k + "_args"] = np.array([frame_index, num_detections])
            fix[k + "_nodes"] = np.array([[tk, n.frame_index, n.id] for tk, trk in enumerate(strack_pool) for n in trk.nodes], np.int64).reshape(-1, 3)
            fix[k + "_ntracks"] = np.array(len(strack_pool))
            return out

        def update_lstm_features(self, tlwh):
            box = np.asarray(tlwh, np.float64).copy()
            saved[2](self, tlwh)
            k = "t%d_mo%d" % (cur["t"], cur["mo"]); cur["mo"] += 1
            fix[k + "_in"] = np.r_[float(self.track_id), float(self.frame_id), box]
            fix[k + "_fut"] = np.stack([np.asarray(self.future_predictions[q]) for q in sorted(self.future_predictions)])
            fix[k + "_h"] = self.hn.reshape(-1).numpy().copy()
        RD.Detector.process, RT.Tracker.get_similarity, RT.STrack.update_lstm_features = process, get_similarity, update_lstm_features
        BaseTrack._count = 0
        det = RD.Detector(opt)
        det.reset_tracking(opt)
        det.img_height, det.img_width = H, W
        afe = det.tracker.model.AFE
        real_ffe = afe.forward_feature_extracter

        def ffe(fm, centers):
            e = real_ffe(fm, centers)
            fix["t%d_centers" % cur["t"]] = centers.detach().cpu().numpy(); fix["t%d_emb" % cur["t"]] = e.detach().cpu().numpy()
            return e
        afe.forward_feature_extracter = ffe
        with torch.no_grad():
            for t in range(T):
                cur.update(t=t, gs=0, mo=0)
                x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(10 + t))
                c = np.array([W / 2.0, H / 2.0], dtype=np.float32)
                meta = {"c": c, "s": np.float32(max(H, W)), "height": H, "width": W, "out_height": H // 4, "out_width": W // 4,
                        "inp_height": H, "inp_width": W, "calib": np.eye(3, 4, dtype=np.float32)}
                batch = lambda v: torch.from_numpy(np.asarray(v)[None])
                targets = det.run({"image": [torch.zeros(H, W, 3)], "images": {1.0: [x]}, "meta": {1.0: {k: batch(v) for k, v in meta.items()}}}, image_info={})
                fix["t%d_tracks" % t] = np.array(sorted([s.track_id] + [float(v) for v in s.tlwh] + [float(s.score)] for s in targets), np.float64).reshape(-1, 6)
                fr = det.tracker.frame_id
                fix["t%d_frame_id" % t] = np.array(fr)
                sims = det.tracker.recorder.all_similarity.get(fr, {})
                fix["t%d_sim_prev" % t] = np.array(sorted(sims), np.int64)
                for p in sims:
                    fix["t%d_sim_%d" % (t, p)] = np.asarray(sims[p], np.float32)
                fix["t%d_boxes" % t] = np.asarray(det.tracker.recorder.all_boxes.get(fr, np.zeros((0, 4))), np.float64)
                fix["t%d_ngs" % t] = np.array(cur["gs"]); fix["t%d_nmo" % t] = np.array(cur["mo"])
        ntr = sum(len(fix["t%d_tracks" % t]) for t in range(T))
        assert ntr >= 12, "the synthetic stream must produce tracks"
    finally:
        RD.Detector.process, RT.Tracker.get_similarity, RT.STrack.update_lstm_features, RT.KalmanFilterLSTM = saved
        torch.cuda.synchronize = real_sync
        os.remove(ck)
    np.savez_compressed(os.path.join(GOLD, "detector_trace_%s.npz" % tag), **fix)
    print("  detector trace [%s]: %d frames, %d track rows, %d seam records" % (tag, T, ntr, len(fix)))


NUSC_INFO_SEED = 3


def nuscenes_image_info():
    """Synthetic calibrated-sensor / ego-pose records of one nuScenes sample (what src/test.py passes as image_info)."""
    from scipy.spatial.transform import Rotation as R
    g = np.random.RandomState(NUSC_INFO_SEED)
    q1, q2 = g.randn(4), g.randn(4)
    return {"trans_matrix": np.concatenate([R.from_rotvec(g.randn(3)).as_matrix(), g.randn(3, 1) * 10], 1).tolist(),
            "cs_record_rot": (q1 / np.linalg.norm(q1)).tolist(), "cs_record_trans": [1.7, 0.0, 1.5],
            "pose_record_rot": (q2 / np.linalg.norm(q2)).tolist(), "pose_record_trans": [411.3, 1180.9, 0.0]}


def nuscenes_trace_state_dict():
    """Synthetic nuScenes net whose detections survive the 0.3 / 0.35 class thresholds of detector.py:222-225 and have positive sizes."""
    sd = dict(O.synth_state_dict("nuscenes"))
    sd["hm.2.weight"] = sd["hm.2.weight"] * 3.0
    sd["hm.2.bias"] = torch.tensor([-1.0, -0.8, -1.2, -0.9, -1.0, -1.1, -0.7, -1.0, -1.0, -1.0])
    sd["dim.2.weight"] = sd["dim.2.weight"] * 0.05
    sd["dim.2.bias"] = torch.tensor([1.6, 1.7, 4.0])
    sd["wh.2.weight"] = sd["wh.2.weight"] * 0.05
    sd["wh.2.bias"] = torch.tensor([6.0, 5.0])
    return sd


def run_detector_trace_nuscenes():
    """BASELINE configs[4]: the reference's OWN nuScenes `Detector.run` (detector.py:112-338: process, post-processing, class thresholds,
    the pyquaternion / nuscenes `Box` chain, per-class NMS, seven per-class Trackers with the LSTM motion model) over 5 synthetic frames,
    with a TRACE of what crosses the seams: process() outputs, the arguments of every `self.tracker[class].update(...)` call
    (detector.py:328-336), the embedding calls, every `update_lstm_features_ddd` and the targets it returns.  pyquaternion and the
    nuScenes devkit are absent here: the reference runs on tests/ref_shims.{Quaternion, Box} -- that part PARITY UNPINNED."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    import importlib
    import ref_shims
    ref_shims.install()
    ref_import.install_stubs(OracleDCN)
    ref_shims.install_detector_stubs()
    argv, sys.argv = sys.argv, ["test.py", "tracking,ddd"]
    try:
        from opts import opts
        from dataset.dataset_factory import dataset_factory
        from utils import tracker as RT
        from utils.basetrack import BaseTrack
        RD = importlib.import_module("detector")
    finally:
        sys.argv = argv
    assert RD.Quaternion is ref_shims.Quaternion and RD.Box is ref_shims.Box
    ck = os.path.join(GOLD, "_trace_ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in nuscenes_trace_state_dict().items()}}, ck)
    H, W, K, T = 64, 96, 12, 5
    opt = opts().parse(["tracking,ddd", "--dataset", "nuscenes", "--gpus", "-1", "--load_model", ck, "--K", str(K),
                        "--input_h", str(H), "--input_w", str(W)])
    opt = opts().update_dataset_info_and_set_heads(opt, dataset_factory[opt.test_dataset])
    opt.lstm = True
    info = nuscenes_image_info()
    names = list(RD.NUSCENES_TRACKING_NAMES)
    calib = np.array([[60.0, 0, W / 2, 0], [0, 60.0, H / 2, 0], [0, 0, 1, 0]], np.float32)
    fix = {"H": H, "W": W, "K": K, "T": T, "seeds": np.arange(10, 10 + T), "calib": calib, "out_thresh": np.array(opt.out_thresh),
           "names": np.array(names)}
    for k, v in info.items():
        fix["info_" + k] = np.asarray(v, np.float64)
    real_sync, torch.cuda.synchronize = torch.cuda.synchronize, (lambda *a, **k: None)
    lsd = O.synth_lstm_state_dict("nuscenes")
    saved = (RD.Detector.process, RT.Tracker.update, RT.STrack.update_lstm_features_ddd, RT.KalmanFilterLSTM, RT.STrack.shared_kalman_lstm)
    cur = {"t": 0, "mo": 0, "emb": 0}
    try:
        class KF(saved[3]):
            def __init__(self, o):
                super().__init__(o)
                self.model.load_state_dict(lsd, strict=True); self.model.eval()
        RT.KalmanFilterLSTM = KF
        RT.STrack.shared_kalman_lstm = KF(opt)

        def process(self, images, *a, **k):
            r = saved[0](self, images, *a, **k)
            for key, v in r[1].items():
                fix["t%d_det_%s" % (cur["t"], key)] = np.asarray(v.detach().cpu() if torch.is_tensor(v) else v)
            return r

        def update(self, results, FeatureMaps, ddd_boxes=None, depths_by_class=None, ddd_org_boxes=None, submission=None, classe=None):
            p = "t%d_cls_%s_" % (cur["t"], classe)
            fix[p + "results"] = np.asarray(results, np.float64).reshape(-1, 5)
            fix[p + "ddd_boxes"] = np.asarray(ddd_boxes, np.float64).reshape(-1, 7)
            fix[p + "depths"] = np.asarray(depths_by_class, np.float64).reshape(-1, 1)
            fix[p + "ddd_org_boxes"] = np.asarray(ddd_org_boxes, np.float64).reshape(-1, 7)
            fix[p + "submission"] = np.asarray(submission, np.float64).reshape(-1, 10)
            return saved[1](self, results, FeatureMaps, ddd_boxes=ddd_boxes, depths_by_class=depths_by_class, ddd_org_boxes=ddd_org_boxes,
                            submission=submission, classe=classe)

        def update_lstm_features_ddd(self, ddd_box):
            box = np.asarray(ddd_box, np.float64).copy()
            saved[2](self, ddd_box)
            k = "t%d_mo%d" % (cur["t"], cur["mo"]); cur["mo"] += 1
            fix[k + "_in"] = np.r_[float(self.track_id), float(self.frame_id), float(names.index(self.classe)), box]
            fix[k + "_fut"] = np.stack([np.asarray(self.future_predictions[q]) for q in sorted(self.future_predictions)])
        RD.Detector.process, RT.Tracker.update, RT.STrack.update_lstm_features_ddd = process, update, update_lstm_features_ddd
        BaseTrack._count = 0
        det = RD.Detector(opt)
        det.reset_tracking(opt)
        det.img_height, det.img_width = H, W
        afe = det.model.AFE                                     # one model behind all seven trackers
        real_ffe = afe.forward_feature_extracter

        def ffe(fm, centers):
            e = real_ffe(fm, centers)
            k = "t%d_emb%d" % (cur["t"], cur["emb"]); cur["emb"] += 1
            fix[k + "_centers"] = centers.detach().cpu().numpy(); fix[k + "_out"] = e.detach().cpu().numpy()
            return e
        afe.forward_feature_extracter = ffe
        ntr = 0
        with torch.no_grad():
            for t in range(T):
                cur.update(t=t, mo=0, emb=0)
                x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(10 + t))
                c = np.array([W / 2.0, H / 2.0], dtype=np.float32)
                meta = {"c": c, "s": np.float32(max(H, W)), "height": H, "width": W, "out_height": H // 4, "out_width": W // 4,
                        "inp_height": H, "inp_width": W, "calib": calib}
                batch = lambda v: torch.from_numpy(np.asarray(v)[None])
                targets = det.run({"image": [torch.zeros(H, W, 3)], "images": {1.0: [x]}, "meta": {1.0: {k: batch(v) for k, v in meta.items()}}},
                                  image_info=info)
                rows = sorted([float(s.track_id), float(names.index(s.classe))] + [float(v) for v in s.tlwh] + [float(s.score)]
                              + [float(v) for v in s.ddd_bbox] + [float(v) for v in s.ddd_submission] for s in targets)
                fix["t%d_targets" % t] = np.array(rows, np.float64).reshape(-1, 24)
                fix["t%d_nmo" % t] = np.array(cur["mo"]); fix["t%d_nemb" % t] = np.array(cur["emb"])
                ntr += len(rows)
        assert ntr >= 15 and len({int(r[1]) for t in range(T) for r in fix["t%d_targets" % t]}) >= 2, "the synthetic stream must produce tracks of several classes"
        assert sum(int(fix["t%d_nmo" % t]) for t in range(T)) >= 10
    finally:
        RD.Detector.process, RT.Tracker.update, RT.STrack.update_lstm_features_ddd, RT.KalmanFilterLSTM, RT.STrack.shared_kalman_lstm = saved
        torch.cuda.synchronize = real_sync
        os.remove(ck)
    np.savez_compressed(os.path.join(GOLD, "detector_trace_nuscenes.npz"), **fix)
    print("  detector trace [nuscenes]: %d frames, %d target rows, %d seam records" % (T, ntr, len(fix)))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if "--only-iou-ddd" in sys.argv:
        run_iou_ddd()
        sys.exit(0)
    if "--only-result-writers" in sys.argv:
        run_result_writers()
        sys.exit(0)
    if "--only-tracker" not in sys.argv:                         # the forward fixtures take a few minutes
        run("mot", 128, 160, "mot_128x160")
        run("mot", 224, 384, "mot_224x384")
        run("nuscenes", 96, 128, "nuscenes_96x128")
        run("kitti_tracking", 96, 320, "kitti_96x320")          # BASELINE config D aspect (1280x384), 3 classes
        run_lstm("mot")
        run_lstm("nuscenes")
        run_convert_detection()
    run_motion("mot")
    run_motion("nuscenes")
    run_track_similarity()
    run_association()
    run_iou_ddd()
    run_result_writers()
    run_postprocess()
    run_detector_trace(lstm=False)
    run_detector_trace(lstm=True)
    run_detector_trace_nuscenes()
    print("golden fixtures written to", os.path.abspath(GOLD))
