"""Vectorised host side of the association step: the cost-matrix builders of the
reference's `utils/matching.py` without their per-track Python loops, and the two third-party entry points
that file imports (`lap.lapjv`, `cython_bbox.bbox_overlaps`; both absent from this image -- `deft_amd/compat/`
exposes the functions below under those module names).

These are <= 100 x 100 float64 problems that feed a host-side assignment; they stay on the host by design (a
launch plus a copy costs more than the arithmetic).  What runs on the device for the association is the part
that touches device data: the track x detection similarity (`deft_amd.tracker.get_similarity`) and the motion
update (`deft_amd.tracker.MotionBank`).

Drop-in: `bind(matching)` replaces `fuse_motion`, `fuse_motion_ddd`, `linear_assignment` and the `bbox_ious`
name inside the reference's `utils.matching` module; the track state machine (Tracker.update) is untouched."""
import numpy as np

chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070, 6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}   # kalman_filter.py:11-21


def bbox_overlaps(boxes, query_boxes):
    """cython_bbox.bbox_overlaps (the Fast R-CNN routine matching.py:4, 71-74 calls): IoU of every box with
    every query box, (x1, y1, x2, y2) with the inclusive-pixel (+1) convention; 0 where they do not overlap.
    float64 [N, K]."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    q = np.asarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + 1
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    area_q = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    inter = iw * ih
    ua = area_b[:, None] + area_q[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where((iw > 0) & (ih > 0), inter / ua, 0.0)


def _host_lib():
    """libdeft_hip.so for its host-side helpers (deft_lapjv, deft_iou3d_matrix: plain C++ on host memory, no GPU involved -- the library
    loads on any box).  No Python re-implementation behind it: one solver, one tie order."""
    from . import hiplib
    return hiplib._lib if hiplib._lib is not None else hiplib.get_lib()


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True, lib=None):
    """`lap.lapjv` (third-party, absent here: parity unpinned) on this repository's own Jonker-Volgenant solver (csrc/assoc.hip,
    `deft_lapjv`): lap's dense algorithm on lap's own extension of the rectangular problem -- with a cost_limit (the only form the
    reference uses, matching.py:48: extend_cost=True, cost_limit=thresh) the (n+m) x (n+m) square with cost_limit/2 in the two
    off-diagonal blocks and 0 in the bottom-right one, so a pair is only matched while it costs less than leaving both unmatched;
    without a limit the zero-padded max(n,m) square (extend_cost=True), or a square matrix is required.  NaN / +inf costs are pairs
    that are never matched.  Ties are broken by the solver's fixed scan order (tests/test_association.py).  Returns (total cost of the
    kept pairs, x, y): x[i] = column of row i or -1, y[j] = row of column j or -1."""
    import ctypes as C
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = cost.shape
    if n != m and not (extend_cost or cost_limit < np.inf):
        raise ValueError("Square cost array expected. If cost is intentionally non-square, pass extend_cost=True.")
    x = np.full(n, -1, dtype=np.int32); y = np.full(m, -1, dtype=np.int32)
    total = C.c_double(0.0)
    if n and m:
        lib = lib if lib is not None else _host_lib()
        rc = lib.cdll.deft_lapjv(C.c_void_p(cost.ctypes.data), n, m, C.c_double(float(cost_limit)), C.c_void_p(x.ctypes.data),
                                 C.c_void_p(y.ctypes.data), C.byref(total))
        if rc != 0:
            raise RuntimeError("deft_lapjv failed (%d): %s" % (rc, lib.cdll.deft_last_error().decode()))
    x, y = x.astype(int), y.astype(int)
    return (float(total.value), x, y) if return_cost else (x, y)


def iou_ddd_distance(a_boxes, b_boxes, lib=None):
    """matching.iou_ddd_distance (matching.py:107-131) on arrays: a_boxes [T, 7], b_boxes [N, 7] (h, w, l, x, y, z, rot_y) ->
    float32 [T, N] = 1 - iou3d(b, a) (csrc/assoc.hip `deft_iou3d_matrix`; the reference loops over pairs in Python and builds a
    scipy ConvexHull per pair)."""
    import ctypes as C
    a = np.ascontiguousarray(a_boxes, dtype=np.float64).reshape(-1, 7)
    b = np.ascontiguousarray(b_boxes, dtype=np.float64).reshape(-1, 7)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if out.size:
        lib = lib if lib is not None else _host_lib()
        rc = lib.cdll.deft_iou3d_matrix(C.c_void_p(a.ctypes.data), a.shape[0], C.c_void_p(b.ctypes.data), b.shape[0], C.c_void_p(out.ctypes.data))
        if rc != 0:
            raise RuntimeError("deft_iou3d_matrix failed (%d): %s" % (rc, lib.cdll.deft_last_error().decode()))
    return out


def linear_assignment(cost_matrix, thresh):
    """matching.py:40-55: (matches [k,2], unmatched rows, unmatched columns)."""
    if cost_matrix.size == 0:
        return np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])), tuple(range(cost_matrix.shape[1]))
    _, x, y = lapjv(cost_matrix, extend_cost=True, cost_limit=thresh)
    rows = np.nonzero(x >= 0)[0]
    return np.stack([rows, x[rows]], 1), np.where(x < 0)[0], np.where(y < 0)[0]


def _maha2(mean2, cov2, meas2):
    """Squared Mahalanobis distance of every measurement to every track, position only: the 2 x 2 Cholesky
    solve of gating_distance (kalman_filter.py:266-275, kalman_filter_lstm.py:92-99) written out, batched over
    tracks.  mean2 [T,2], cov2 [T,2,2], meas2 [N,2] -> [T,N]."""
    with np.errstate(invalid="ignore", divide="ignore"):
        l00 = np.sqrt(cov2[:, 0, 0])
        l10 = cov2[:, 1, 0] / l00
        l11 = np.sqrt(cov2[:, 1, 1] - l10 * l10)
    if not (np.all(np.isfinite(l00)) and np.all(np.isfinite(l11)) and np.all(l00 > 0) and np.all(l11 > 0)):
        raise np.linalg.LinAlgError("Matrix is not positive definite")       # what np.linalg.cholesky raises
    z0 = (meas2[None, :, 0] - mean2[:, None, 0]) / l00[:, None]                  # ([T, N] planes: no [T, N, 2] temporary)
    z1 = ((meas2[None, :, 1] - mean2[:, None, 1]) - l10[:, None] * z0) / l11[:, None]
    return z0 * z0 + z1 * z1


def fuse_motion(kf, cost_matrix, tracks, detections, frame_id, use_lstm=True, only_position=True, lambda_=0.9):
    """matching.py:311-371 for all tracks at once (in place, like the reference).  Kalman tracks and LSTM tracks
    with >= 300 observations: gate at 5 * chi2inv95 on the squared Mahalanobis distance, add 0.05*(1-lambda)
    of it; younger LSTM tracks: the reference's "gaussian" distance, which on this 2-D position-only path is
    identically 0 (kalman_filter_lstm.py:87-91 slices an already 2-wide vector with [3:-1]), so those rows are
    just scaled by lambda."""
    if cost_matrix.size == 0:
        return cost_matrix
    if not only_position:                                  # never used by the reference's tracker; keep its loop
        raise NotImplementedError("fuse_motion: only_position=False is not on the tracker's path")
    thr = chi2inv95[2]
    meas = np.asarray([det.to_xyah() for det in detections])[:, :2]
    if use_lstm:
        maha = np.array([len(t.observations) >= 300 for t in tracks])
        means = [t.prediction_at_frame(frame_id) for t in tracks]
    else:
        maha = np.ones(len(tracks), dtype=bool)
        means = [t.mean for t in tracks]
    if maha.any():
        idx = np.nonzero(maha)[0]
        mean2 = np.asarray([np.asarray(means[i], dtype=np.float64)[:2] for i in idx])
        cov2 = np.asarray([np.asarray(tracks[i].covariance, dtype=np.float64)[:2, :2] for i in idx])
        g = _maha2(mean2, cov2, meas)
        rows = cost_matrix[idx]
        rows[g > 5.0 * thr] = np.inf
        cost_matrix[idx] = lambda_ * rows + 0.05 * (1 - lambda_) * g
    if (~maha).any():
        cost_matrix[~maha] = lambda_ * cost_matrix[~maha]
    return cost_matrix


def ddd_metric_of(kf):
    """Which "gaussian" distance the filter object `kf` stands for in fuse_motion_ddd: "centre" (KalmanFilterLSTM.gating_distance,
    kalman_filter_lstm.py:92-95) or "squared7" (KalmanFilter.gating_distance, kalman_filter.py:271-273).  Decided by an explicit
    `ddd_metric` attribute when the object has one, else by the class names in its MRO (subclasses and this package's mirrors of the two
    reference classes resolve like their base); anything else is an error -- a silently wrong metric would change the nuScenes gating."""
    m = getattr(kf, "ddd_metric", None)
    if m is not None:
        if m not in ("centre", "squared7"):
            raise ValueError("fuse_motion_ddd: unknown ddd_metric %r" % (m,))
        return m
    names = [c.__name__ for c in type(kf).__mro__]
    if "KalmanFilterLSTM" in names:
        return "centre"
    if "KalmanFilter" in names:
        return "squared7"
    raise TypeError("fuse_motion_ddd: cannot tell the 3-D gating metric of a %s (neither KalmanFilter nor KalmanFilterLSTM, no ddd_metric attribute)"
                    % type(kf).__name__)


def fuse_motion_ddd(kf, cost_matrix, tracks, detections, frame_id, use_lstm=True, only_position=False, lambda_=0.9,
                    use_prediction=False, classe_name=None):
    """matching.py:374-415 for all tracks at once: the filter's "gaussian" distance between track box and detection box -- with the LSTM
    motion model the distance between the 3-D centres (components 3..5 of the (h,w,l,x,y,z,rot) boxes, kalman_filter_lstm.py:92-95);
    with the plain KalmanFilter (opt.lstm off: tracker.py:652) the SQUARED distance over all seven components (kalman_filter.py:271-273)
    -- gate at max(0.2 * depth, 5 pedestrians / 10 others), add 0.001 of it."""
    if cost_matrix.size == 0:
        return cost_matrix
    if only_position:
        raise NotImplementedError("fuse_motion_ddd: only_position=True is not on the tracker's path")
    meas = np.asarray([det.ddd_bbox for det in detections], dtype=np.float64)
    boxes = np.asarray([(t.ddd_prediction_at_frame(frame_id) if use_prediction else t.ddd_bbox) for t in tracks], dtype=np.float64)
    metric = ddd_metric_of(kf)
    if metric == "squared7":                                 # the reference's constant-velocity filter class, used when opt.lstm is off
        d = meas[None, :, :] - boxes[:, None, :]
        g = np.sum(d * d, axis=2)
    else:
        d = meas[None, :, 3:-1] - boxes[:, None, 3:-1]
        g = np.sqrt(np.sum(d * d, axis=2))
    floor = 5 if classe_name == "pedestrian" else 10
    thr = np.maximum(0.2 * np.asarray([t.depth for t in tracks], dtype=np.float64), floor)
    cost_matrix[g > thr[:, None]] = np.inf
    cost_matrix[:] = lambda_ * cost_matrix + 0.001 * g
    return cost_matrix


def bind(matching):
    """Swap the loops of the reference's `utils.matching` module for the functions above.  Returns undo()."""
    names = {"fuse_motion": fuse_motion, "fuse_motion_ddd": fuse_motion_ddd, "linear_assignment": linear_assignment,
             "bbox_ious": bbox_overlaps}
    saved = {k: getattr(matching, k) for k in names}
    for k, v in names.items():
        setattr(matching, k, v)

    def undo():
        for k, v in saved.items():
            setattr(matching, k, v)
    return undo
