"""The 2-D tracking loop of DEFT (MOT17 / KITTI) on the accelerated path, frame in -> tracks out, without the reference tree:
`Tracker.update` + `STrack` + the DeepSORT Kalman filter (utils/tracker.py:626-1056, 140-420; utils/tracking_utils/kalman_filter.py)
re-expressed over this package's device forms --

  * detections -> box centres -> embeddings: `model.AFE.forward_feature_extracter` (AfeSeam; tracker.py:807-826);
  * affinity against every stored frame: `deft_amd.tracker.FeatureRecorder` (one launch chain per frame, tracker.py:59-90);
  * tracks x detections similarity: `deft_amd.tracker.get_similarity` (device-side decay + median, tracker.py:219-252, 663-688);
  * motion gating / assignment / IoU: `deft_amd.association` (matching.py:40-104, 311-371), all tracks at once;
  * the Kalman filter batched over the track pool (multi_predict / update as array operations instead of one scipy Cholesky per track).

SURVEY.md 8(f) rank 1: "once kernels are fast, the Python/numpy loops and lap dominate" -- this is the association step as arrays.
Semantics kept exactly as the reference runs them (checked frame by frame against the reference's own Tracker in
tests/test_mot_tracker.py, ids and boxes):
  * an unmatched track is never marked Lost: it stays in `tracked_stracks` (state Tracked) until it has not been seen for more than
    `max_time_lost` frames, then it is removed (tracker.py:1006-1010; `lost_stracks` stays empty);
  * `is_activated` only becomes true on frame 1 or at the first match (tracker.py:209-217, 263-277); every matched track and every
    new detection is returned (tracker.py:1012-1018), activated or not;
  * KITTI: a second, similarity-only association for the detections the first one left over, and tracks are kept as IoU
    candidates while seen within the last 6 frames (tracker.py:956-995); MOT: tracks in state Tracked;
  * track ids come from one process-wide counter that is never reset between videos (basetrack.py:18, 40-42: `BaseTrack._count`):
    `TrackIds` below, shared by every tracker of the process.
Scope: the 2-D datasets with the Kalman motion model (the configuration of BASELINE configs[1] / [2]); `--lstm` and nuScenes keep
using the reference's Tracker with `deft_amd.tracker.accelerate` bound.
"""
import numpy as np

from . import association as A
from . import tracker as DT

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3          # basetrack.py:11-15


class TrackIds:
    """basetrack.py:18, 40-42."""
    count = 0

    @classmethod
    def next_id(cls):
        cls.count += 1
        return cls.count


class Node:
    """tracker.py:28-43: which detection of which frame."""
    __slots__ = ("frame_index", "id")

    def __init__(self, frame_index, id):
        self.frame_index, self.id = frame_index, id


# ---------------------------------------------------------------------------------------------------------------------
# DeepSORT Kalman filter, batched (utils/tracking_utils/kalman_filter.py:24-275)
# ---------------------------------------------------------------------------------------------------------------------
_F = np.eye(8)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0                            # kalman_filter.py:41-44 (dt = 1)
_H = np.eye(4, 8)
_SP, _SV = 1.0 / 20, 1.0 / 160                      # kalman_filter.py:50-51


def kf_initiate(xyah):
    """kalman_filter.py:53-88."""
    mean = np.r_[xyah, np.zeros(4)]
    h = xyah[3]
    std = [2 * _SP * h, 2 * _SP * h, 1e-2, 2 * _SP * h, 10 * _SV * h, 10 * _SV * h, 1e-5, 10 * _SV * h]
    return mean, np.diag(np.square(std))


def kf_multi_predict(mean, cov):
    """kalman_filter.py:165-205: mean [T, 8], cov [T, 8, 8]."""
    h = mean[:, 3]
    one = np.ones_like(h)
    sqr = np.square(np.stack([_SP * h, _SP * h, 1e-2 * one, _SP * h, _SV * h, _SV * h, 1e-5 * one, _SV * h], 1))
    mean = np.dot(mean, _F.T)
    left = np.dot(_F, cov).transpose((1, 0, 2))
    cov = np.dot(left, _F.T)
    idx = np.arange(8)
    cov[:, idx, idx] += sqr
    return mean, cov


def kf_multi_update(mean, cov, meas):
    """kalman_filter.py:207-240 for T tracks at once: mean [T, 8], cov [T, 8, 8], meas [T, 4] (x, y, a, h)."""
    h = mean[:, 3]
    std = np.stack([_SP * h, _SP * h, 1e-1 * np.ones_like(h), _SP * h], 1)
    pm = mean[:, :4]                                                      # H . mean
    pc = cov[:, :4, :4].copy()                                            # H P H^T
    idx = np.arange(4)
    pc[:, idx, idx] += np.square(std)
    # K = P H^T S^-1  (the reference solves with a Cholesky factor of S; S is SPD, the solve below gives the same K to round-off)
    gain = np.linalg.solve(pc, cov[:, :4, :]).transpose(0, 2, 1)          # [T, 8, 4]
    innov = meas - pm
    new_mean = mean + np.einsum("ti,tji->tj", innov, gain)
    new_cov = cov - np.einsum("tij,tjk,tlk->til", gain, pc, gain)
    return new_mean, new_cov


def tlbr_to_tlwh(tlbr):
    """STrack.tlbr_to_tlwh (tracker.py:600-604): in the dtype of its argument -- the tracker hands float32 rows, so the width / height
    are float32 differences (widened afterwards by STrack.__init__)."""
    r = np.asarray(tlbr).copy()
    r[2:] -= r[:2]
    return r


def tlwh_to_xyah(tlwh):
    r = np.asarray(tlwh, dtype=float).copy()
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


class Track:
    """STrack (tracker.py:140-306) for the Kalman configuration: the fields `Tracker.update`, the result writers (src/test.py:226-258)
    and deft_amd.association read."""
    __slots__ = ("_tlwh", "_xyah", "_tlbr", "mean", "covariance", "is_activated", "score", "tracklet_len", "nodes", "track_id", "state",
                 "frame_id", "start_frame")

    def __init__(self, tlwh, score, node, xyah=None, tlbr=None):
        """xyah / tlbr: the detection's (x, y, a, h) and corner forms when the caller has computed them for the whole frame at once (the
        same float64 expressions as to_xyah() / .tlbr below, row by row); they describe the detection, i.e. hold until a filter state exists."""
        self._tlwh = np.asarray(tlwh, dtype=float)
        self._xyah, self._tlbr = xyah, tlbr
        self.mean = self.covariance = None
        self.is_activated = False
        self.score = score
        self.tracklet_len = 0
        self.nodes = [node]
        self.track_id, self.state, self.frame_id, self.start_frame = 0, NEW, 0, 0

    end_frame = property(lambda self: self.frame_id)

    @property
    def tlwh(self):
        if self.mean is None:
            return self._tlwh.copy()
        r = self.mean[:4].copy()
        r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):
        if self.mean is None and self._tlbr is not None:
            return self._tlbr.copy()
        r = self.tlwh
        r[2:] += r[:2]
        return r

    def to_xyah(self):
        if self.mean is None and self._xyah is not None:
            return self._xyah.copy()
        return tlwh_to_xyah(self.tlwh)

    def __repr__(self):
        return "OT_{}_({}-{})".format(self.track_id, self.start_frame, self.end_frame)


class Tracker2D:
    def __init__(self, opt, model, h=100, w=100, frame_rate=10):
        """opt: dataset ("mot" | "kitti_tracking"), track_buffer, max_object.  model: carries `.AFE` (deft_amd.integrate.AfeSeam).
        h, w: the image size detection centres are normalised with (tracker.py:817-820; 100 until reset_tracking passes the real one)."""
        assert opt.dataset in ("mot", "kitti_tracking") and not getattr(opt, "lstm", False), \
            "Tracker2D is the Kalman configuration of the 2-D datasets; --lstm / nuScenes: the reference Tracker + deft_amd.tracker.accelerate"
        self.opt, self.dataset, self.model = opt, opt.dataset, model
        self.img_height, self.img_width = h, w
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self.frame_id = 0
        self.max_time_lost = int(frame_rate / 30.0 * getattr(opt, "track_buffer", 30))          # tracker.py:648-649
        self.recorder = DT.FeatureRecorder(opt.dataset)
        self.det_thresh = 0.0
        self.use_lstm = False

    # deft_amd.tracker.get_similarity reads .recorder / .dataset / .model.AFE
    def get_similarity(self, frame_index, pool, num_detections, selected=None):
        return DT.get_similarity(self, frame_index, pool, num_detections, selected)

    def _rows(self, results):
        if self.dataset == "kitti_tracking":
            rows = [np.asarray(d["bbox"]).tolist() + [d["score"]] for d in results if d["class"] == 2]          # tracker.py:790-797
        else:
            rows = [np.asarray(d["bbox"]).tolist() + [d["score"]] for d in results]
        return np.array(rows, np.float32)

    @staticmethod
    def _match(cost, thresh, tracks, detections, frame_id, matched_out, activated):
        matches, u_t, u_d = A.linear_assignment(cost, thresh)
        if len(matches):
            ti, di = matches[:, 0], matches[:, 1]
            sel = [tracks[i] for i in ti]
            mean = np.stack([t.mean for t in sel]); cov = np.stack([t.covariance for t in sel])
            meas = np.stack([detections[j].to_xyah() for j in di])
            mean, cov = kf_multi_update(mean, cov, meas)                  # STrack.update / re_activate, tracker.py:235-293
            for k, (t, j) in enumerate(zip(sel, di)):
                d = detections[j]
                if t.state == TRACKED:
                    t.tracklet_len += 1
                    t.score = d.score
                    activated.append(t)
                else:
                    t.tracklet_len = 0
                t.frame_id, t.state, t.is_activated = frame_id, TRACKED, True
                t.nodes.append(d.nodes[-1])
                t.mean, t.covariance = mean[k], cov[k]
                matched_out.append(t)
        return u_t, u_d

    lazy_blocks = True             # score the new frame only against the stored frames the pool's selected nodes live in (FeatureRecorder.update)
    after_device_work = None       # set by a caller (Detector.run's lookahead): called ONCE per update(), as soon as the frame's last
    #                                device-dependent step has returned -- what follows is host work, and the GPU is free for the next frame

    def _device_done(self):
        cb, self.after_device_work = self.after_device_work, None
        if cb is not None:
            cb()

    def update(self, results, FeatureMaps):
        """tracker.py:726-1056 (2-D branch, Kalman).  results: the frame's post-processed detections ({"bbox" tlbr, "score", "class"});
        FeatureMaps: the 13 maps of the frame.  Returns the tracks matched or started in this frame."""
        self.frame_id += 1
        fid = self.frame_id
        activated, removed, output = [], [], []
        dets = self._rows(results)
        pool = list(self.tracked_stracks) + [t for t in self.lost_stracks if t.track_id not in {x.track_id for x in self.tracked_stracks}]
        sel = {id(t): DT.select_nodes(t.nodes, fid, self.dataset) for t in pool}      # once per track and frame: block selection + both associations
        if len(dets) > 0:
            tlwh32 = dets[:, :4].copy()                                   # STrack.tlbr_to_tlwh: float32 differences, widened afterwards
            tlwh32[:, 2:] -= tlwh32[:, :2]
            tlwh = tlwh32.astype(float)
            xyah = tlwh.copy()                                            # to_xyah / tlbr of every detection at once (same float64 expressions)
            xyah[:, :2] += xyah[:, 2:] / 2
            xyah[:, 2] /= xyah[:, 3]
            tlbr = tlwh.copy()
            tlbr[:, 2:] += tlbr[:, :2]
            detections = [Track(tlwh[i], dets[i, 4], Node(fid, i), xyah[i], tlbr[i]) for i in range(len(dets))]
            org = np.copy(dets[:, :4])
            d = np.array(org, dtype=np.float64)                           # convert_detection, image.py:391-412
            d[:, 2] -= d[:, 0]; d[:, 3] -= d[:, 1]
            d[:, 0] /= self.img_width; d[:, 2] /= self.img_width; d[:, 1] /= self.img_height; d[:, 3] /= self.img_height
            import torch
            centers = torch.from_numpy(((2 * d[:, 0:2] + d[:, 2:4]) - 1.0).astype(float)).float().view(1, -1, 1, 1, 2)
            feats = self.model.AFE.forward_feature_extracter(FeatureMaps, centers)
            # only the stored frames the association below can read: those holding one of the last few nodes of a pooled track
            needed = {n.frame_index for nodes in sel.values() for n in nodes} if self.lazy_blocks else None
            self.recorder.update(self.model, fid, feats.data, org, needed=needed)
        else:
            detections = []
        if pool:                                                           # STrack.multi_predict, tracker.py:193-207
            mean = np.stack([t.mean for t in pool]); cov = np.stack([t.covariance for t in pool])
            if any(t.state != TRACKED for t in pool):
                mean = mean.copy()
                for i, t in enumerate(pool):
                    if t.state != TRACKED:
                        mean[i, 7] = 0
            mean, cov = kf_multi_predict(mean, cov)
            for i, t in enumerate(pool):
                t.mean, t.covariance = mean[i], cov[i]
        nd0 = len(detections)
        # ---- first association: embedding similarity fused with the Kalman gate (tracker.py:879-915) ----
        dists = np.zeros((len(pool), nd0), dtype=float)
        if dists.size:
            dists = 1 - self.get_similarity(fid, pool, nd0, sel)[:, :-1]
        if self.dataset != "kitti_tracking":
            self._device_done()
        if dists.size:
            # matching.fuse_motion (matching.py:311-371; Kalman branch, position only) on the arrays at hand -- the predicted means / covariances
            # stacked above and the frame's (x, y, a, h) rows -- instead of collecting them again track by track: the same expressions as
            # association.fuse_motion (gate at 5 * chi2inv95[2] on the squared Mahalanobis distance, then 0.9 * cost + 0.05 * 0.1 * distance)
            lam = 0.9
            g = A._maha2(mean[:, :2], cov[:, :2, :2], xyah[:, :2])
            dists[g > 5.0 * A.chi2inv95[2]] = np.inf
            dists = lam * dists + 0.05 * (1 - lam) * g
        u_track, u_det2 = self._match(dists, 0.9, pool, detections, fid, output, activated)
        r_tracked = [pool[i] for i in u_track]
        detections = [detections[i] for i in u_det2]
        if self.dataset == "kitti_tracking" and detections:                # second, similarity-only association (tracker.py:956-980)
            dists = self.get_similarity(fid, r_tracked, nd0, sel)
            if dists.size:
                dists = 1 - dists[:, :-1][:, u_det2]
                u_track, u_det = self._match(dists, 0.9, r_tracked, detections, fid, output, activated)
                detections = [detections[i] for i in u_det]
                pool = r_tracked
        self._device_done()
        if self.dataset == "kitti_tracking":
            r_tracked = [pool[i] for i in u_track if abs(fid - pool[i].frame_id) < 6]            # tracker.py:982-990
        else:
            r_tracked = [pool[i] for i in u_track if pool[i].state == TRACKED]                   # tracker.py:992-997
        # ---- IoU association of what is left (tracker.py:1008-1030) ----
        if r_tracked and detections:
            cost = 1 - A.bbox_overlaps(np.ascontiguousarray([t.tlbr for t in r_tracked]), np.ascontiguousarray([t.tlbr for t in detections]))
        else:
            cost = np.zeros((len(r_tracked), len(detections)), dtype=float)
        u_track, u_det = self._match(cost, 0.9, r_tracked, detections, fid, output, activated)
        for it in u_track:
            t = r_tracked[it]
            if fid - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        for j in u_det:                                                    # new tracklets (tracker.py:1012-1018, STrack.activate :209-233)
            t = detections[j]
            output.append(t)
            if t.score < self.det_thresh:
                continue
            t.track_id = TrackIds.next_id()
            t.tracklet_len, t.state = 0, TRACKED
            if fid == 1:
                t.is_activated = True
            t.frame_id = t.start_frame = fid
            t.mean, t.covariance = kf_initiate(tlwh_to_xyah(t._tlwh))
            activated.append(t)
        # ---- state update (tracker.py:1019-1042; lost_stracks stays empty in this tracker: nothing is ever marked Lost) ----
        for t in self.lost_stracks:
            if fid - t.end_frame > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        tracked = [t for t in self.tracked_stracks if t.state == TRACKED]
        seen = {t.track_id for t in tracked}
        for t in activated:                                                # joint_stracks
            if t.track_id not in seen:
                seen.add(t.track_id)
                tracked.append(t)
        self.tracked_stracks = tracked
        self.lost_stracks = [t for t in self.lost_stracks if t.track_id not in seen and t.state != REMOVED]
        self.removed_stracks.extend(removed)
        return output
