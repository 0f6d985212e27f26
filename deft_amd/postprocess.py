"""Post-processing of the decoded detections on the fused path: what the reference does between
`Detector.process` and `Tracker.update`, vectorised over the K detections of a frame instead of a Python loop per detection.

  generic_post_process   utils/post_process.py:29-112 -- inverse affine (network output grid -> original image pixels) of
                         centres / boxes / tracking offsets, observation angle from the two rotation bins, amodal centre,
                         depth un-projection + rotation_y (utils/ddd_utils.py:128-165, 162-169);
  merge_outputs          detector.py:577-583 (`score > out_thresh`);
  nuscenes_frame         the nuScenes branch of `Detector.run` (detector.py:200-338): per-class score filters, size order,
                         camera -> ego -> global quaternion chain, submission boxes, greedy NMS with the reference's
                         `sorted(set(keep))` quirk (the zero-initialised `keep` always contains index 0, ddd_utils.py:193-245)
                         -> the argument lists of `self.tracker[class_name].update(...)`;
  greedy_nms             utils/ddd_utils.py:178-245.

Host code on purpose: <= 100 rows of float32 arithmetic per frame feeding host-side association -- a launch + copy would cost
more than the arithmetic (same call as deft_amd/association.py).  Float32 where the reference computes in float32, so the
values are the reference's to the last bit on the pinned parts (tests/golden/postprocess.npz, written from the reference's own
functions by oracle/make_golden.py).  PARITY UNPINNED for two third-party pieces that are absent from /root/reference and from
this image: `cv2.getAffineTransform` (a 3-point affine solve; restated as a float64 linear solve) and `pyquaternion.Quaternion`
+ `nuscenes.utils.data_classes.Box` (restated below from their published semantics, cross-checked against
scipy.spatial.transform.Rotation in tests/test_postprocess.py).
"""
import numpy as np

NUSCENES_TRACKING_NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck"]                 # detector.py:39-47
NUSCENES_CLASS_NAME = ["car", "truck", "bus", "trailer", "construction_vehicle", "pedestrian", "motorcycle", "bicycle",
                       "traffic_cone", "barrier"]                                                                      # detector.py:49-60


# ---------------------------------------------------------------------------------------------------------------------
# affine (utils/image.py:42-72 with rot = 0, inv = 1)
# ---------------------------------------------------------------------------------------------------------------------
def inverse_affine(c, s, w, h):
    """get_affine_transform(c, s, 0, (w, h), inv=1).astype(float32): the 2x3 matrix from the (w x h) network output grid back
    to original-image pixels.  The three point pairs are the reference's; the solve stands in for cv2.getAffineTransform."""
    c = np.asarray(c, np.float32).reshape(2)
    s = np.asarray(s, np.float32).reshape(-1)
    s = np.array([s[0], s[0]], np.float32) if s.size == 1 else s[:2]
    src = np.zeros((3, 2), np.float32); dst = np.zeros((3, 2), np.float32)
    src[0] = c
    src[1] = c + np.array([0.0, s[0] * -0.5], np.float32)
    dst[0] = [w * 0.5, h * 0.5]
    dst[1] = np.array([w * 0.5, h * 0.5], np.float32) + np.array([0, w * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    A = np.concatenate([dst.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, src.astype(np.float64)).T.astype(np.float32)


def _apply(coords, trans):
    """transform_preds_with_trans (utils/image.py:25-31) for [n,2] points, float32."""
    t = np.ones((coords.shape[0], 3), np.float32)
    t[:, :2] = coords
    return np.dot(trans, t.T).T[:, :2]


def generic_post_process(dets, c, s, h, w, out_thresh, calib=None):
    """post_process.py:29-112 for one frame (`dets`: generic_decode's arrays with the batch dimension, as
    deft_amd.detector.Detector.process returns them).  Rows are kept up to the first score below out_thresh
    (the reference's `break`; scores arrive sorted).  -> dict of arrays over the n kept rows:
    score, class (1-based), ct [n,2], bbox [n,4], tracking [n,2]; with 3-D heads also dep, dim [n,3], alpha, loc [n,3], rot_y."""
    sc = np.asarray(dets["scores"][0])
    below = np.nonzero(sc < out_thresh)[0]
    n = int(below[0]) if below.size else sc.shape[0]
    trans = inverse_affine(c, s, w, h)
    out = {"score": sc[:n].copy(), "class": np.asarray(dets["clses"][0][:n]).astype(np.int64) + 1}
    cts = np.asarray(dets["cts"][0][:n], np.float32)
    out["ct"] = _apply(cts, trans)
    if "tracking" in dets:
        out["tracking"] = _apply(np.asarray(dets["tracking"][0][:n], np.float32) + cts, trans) - out["ct"]
    if "bboxes" in dets:
        bb = np.asarray(dets["bboxes"][0][:n], np.float32)
        out["bbox"] = _apply(bb.reshape(-1, 2), trans).reshape(n, 4)
    if "dep" in dets:
        out["dep"] = np.asarray(dets["dep"][0][:n], np.float32).reshape(n, -1)
    if "dim" in dets:
        out["dim"] = np.asarray(dets["dim"][0][:n], np.float32)
    if "rot" in dets:
        rot = np.asarray(dets["rot"][0][:n], np.float32)
        idx = rot[:, 1] > rot[:, 5]                                                  # get_alpha, post_process.py:19-26
        a1 = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
        a2 = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
        out["alpha"] = a1 * idx + a2 * (1 - idx)
    if "rot" in dets and "dep" in dets and "dim" in dets:
        if "amodel_offset" in dets:
            ct_out = bb.reshape(n, 2, 2).mean(axis=1) + np.asarray(dets["amodel_offset"][0][:n], np.float32)
            ct = _apply(ct_out, trans)
        else:
            ct = np.stack([(out["bbox"][:, 0] + out["bbox"][:, 2]) / 2, (out["bbox"][:, 1] + out["bbox"][:, 3]) / 2], 1)
        out["ct"] = ct
        P = np.asarray(calib, np.float32)
        depth = out["dep"][:, 0]
        z = depth - P[2, 3]                                                          # unproject_2d_to_3d, ddd_utils.py:128-137
        x = (ct[:, 0] * depth - P[0, 3] - P[0, 2] * z) / P[0, 0]
        y = (ct[:, 1] * depth - P[1, 3] - P[1, 2] * z) / P[1, 1]
        loc = np.stack([x, y, z], 1).astype(np.float32)
        loc[:, 1] += out["dim"][:, 0] / 2                                            # ddd2locrot, ddd_utils.py:162-168
        ry = out["alpha"] + np.arctan2(ct[:, 0] - P[0, 2], P[0, 0])                  # alpha2rot_y, ddd_utils.py:140-151
        ry = np.where(ry > np.pi, ry - 2 * np.pi, ry)
        ry = np.where(ry < -np.pi, ry + 2 * np.pi, ry)
        out["loc"], out["rot_y"] = loc, ry.astype(np.float32)
    return out


def merge_outputs(post, out_thresh):
    """detector.py:577-583: rows with score > out_thresh (strict, unlike the `break` above)."""
    keep = post["score"] > out_thresh
    return {k: v[keep] for k, v in post.items()}


class ResultList(list):
    """The reference's list of per-detection dicts, with the arrays it was built from riding along (`post`: the same rows in the same order, or
    None) -- ArrayTracker.detections_as_arrays reads those instead of parsing 100 dicts back into the arrays they came from."""
    __slots__ = ("post",)

    def arrays(self):
        """`post` if it still describes this list (nobody added or removed rows), else None."""
        post = getattr(self, "post", None)
        return post if post is not None and post["score"].shape[0] == len(self) else None


def as_result_list(post):
    """The list of per-detection dicts `Tracker.update` / the result writers consume (test.py:220-258)."""
    n = post["score"].shape[0]
    out = ResultList({k: (int(v[i]) if k == "class" else v[i]) for k, v in post.items()} for i in range(n))
    out.post = post
    return out


# ---------------------------------------------------------------------------------------------------------------------
# quaternions (w, x, y, z), pyquaternion / nuscenes Box semantics -- UNPINNED restatement
# ---------------------------------------------------------------------------------------------------------------------
def q_axis_angle(axis, angle):
    """Quaternion(axis=axis, angle=angle): unit axis, (cos(angle/2), sin(angle/2) * axis).  angle may be an array [n]."""
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    half = np.asarray(angle, np.float64) / 2.0
    return np.concatenate([np.cos(half)[..., None], np.sin(half)[..., None] * axis], -1)


def q_mul(a, b):
    """Hamilton product a * b (broadcasting over leading dimensions)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], -1)


def q_matrix(q):
    """Rotation matrix of a (normalised) quaternion, [..., 3, 3]."""
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def q_angle_axis(q):
    """pyquaternion's `.angle` (2*atan2(|v|, w) wrapped to (-pi, pi]) and `.axis` (v / |v|, zeros when |v| ~ 0)."""
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    vn = np.linalg.norm(q[..., 1:], axis=-1)
    th = 2.0 * np.arctan2(vn, q[..., 0])
    ang = ((th + np.pi) % (2 * np.pi)) - np.pi
    ang = np.where(ang == -np.pi, np.pi, ang)
    axis = np.where(vn[..., None] < 1e-17, 0.0, q[..., 1:] / np.maximum(vn[..., None], 1e-300))
    return ang, axis


# ---------------------------------------------------------------------------------------------------------------------
# greedy NMS (utils/ddd_utils.py:178-245) and the nuScenes branch of Detector.run
# ---------------------------------------------------------------------------------------------------------------------
def greedy_nms(boxes, scores, overlap=0.93, top_k=200, lib=None):
    """-> (keep, count): `keep` is the reference's zero-initialised index vector of FULL length with the kept indices in its
    first `count` slots (highest score first).  lib: a bound HipLib -- the same loop as one native host call (assoc.hip deft_greedy_nms);
    this numpy form stays as its cross-check."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4); scores = np.asarray(scores, np.float64).reshape(-1)
    keep = np.zeros(scores.shape[0], np.int64)
    if boxes.size == 0:
        return keep, 0
    if lib is not None and "deft_greedy_nms" in getattr(lib, "_fn", ()):
        import ctypes as C
        boxes, scores = np.ascontiguousarray(boxes), np.ascontiguousarray(scores)
        cnt = C.c_int(0)
        rc = lib._fn["deft_greedy_nms"](C.c_void_p(boxes.ctypes.data), C.c_void_p(scores.ctypes.data), scores.shape[0], float(overlap), int(top_k),
                                        C.c_void_p(keep.ctypes.data), C.byref(cnt))
        if rc != 0:
            raise RuntimeError("deft_greedy_nms failed (%d): %s" % (rc, lib.last_error()))
        return keep, int(cnt.value)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    idx = np.argsort(scores, kind="stable")[-top_k:]               # ascending; torch.sort's order among equal scores is unspecified
    count = 0
    while idx.size > 0:
        i = idx[-1]
        keep[count] = i; count += 1
        if idx.size == 1:
            break
        idx = idx[:-1]
        w = np.clip(np.minimum(x2[idx], x2[i]) - np.maximum(x1[idx], x1[i]), 0.0, None)
        h = np.clip(np.minimum(y2[idx], y2[i]) - np.maximum(y1[idx], y1[i]), 0.0, None)
        inter = w * h
        union = (area[idx] - inter) + area[i]
        idx = idx[(inter / union) <= overlap]
    return keep, count


def nuscenes_frame(post, image_info, nms=True, lib=None):
    """The nuScenes branch of `Detector.run` (detector.py:200-338) on the merged post-processed detections `post`
    (arrays: score, class, bbox, dim, loc, rot_y).  image_info: trans_matrix [3|4 x 4], cs_record_rot / pose_record_rot
    (w,x,y,z), cs_record_trans / pose_record_trans.  -> {class_name: dict(results [m,5], ddd_boxes [m,7], depths [m,1],
    ddd_org_boxes [m,7], submission [m,10])} = the arguments of `self.tracker[class_name].update(results, FeatureMaps,
    ddd_boxes=, depths_by_class=, ddd_org_boxes=, submission=, classe=class_name)`."""
    trans_matrix = np.array(image_info["trans_matrix"], np.float32)
    cls = np.asarray(post["class"]).astype(np.int64)
    score = np.asarray(post["score"])
    names = np.array(NUSCENES_CLASS_NAME, dtype=object)[cls - 1]
    ok = np.isin(names, NUSCENES_TRACKING_NAMES) & ~(score < 0.3) & ~((names == "pedestrian") & (score < 0.35))
    dim = np.asarray(post["dim"], np.float32); loc = np.asarray(post["loc"], np.float32)
    rot_y = np.asarray(post["rot_y"], np.float64).reshape(-1)
    n = score.shape[0]
    size = np.stack([dim[:, 1], dim[:, 2], dim[:, 0]], 1).astype(np.float64)            # [float(dim[1]), float(dim[2]), float(dim[0])]
    # translation_submission1 = trans_matrix . [x, y - size[2], z, 1] in float32 (detector.py:233-239)
    hom = np.stack([loc[:, 0], (loc[:, 1].astype(np.float64) - size[:, 2]).astype(np.float32), loc[:, 2], np.ones(n, np.float32)], 1).astype(np.float32)
    sub_t = np.dot(trans_matrix, hom.T).T[:, :3].astype(np.float64)
    # Box(loc, size, Quaternion(axis=[0,1,0], angle=rot_y)); translate(0, -wlh[2]/2, 0); rotate(cs); translate(cs); rotate(pose); translate(pose)
    center = loc.astype(np.float64).copy()
    orient = q_axis_angle([0, 1, 0], rot_y)
    center[:, 1] += -size[:, 2] / 2
    for rq, tr in ((image_info["cs_record_rot"], image_info["cs_record_trans"]), (image_info["pose_record_rot"], image_info["pose_record_trans"])):
        rq = np.asarray(rq, np.float64)
        center = center @ q_matrix(rq).T
        orient = q_mul(rq, orient)
        center = center + np.asarray(tr, np.float64)
    ang, axis = q_angle_axis(orient)
    angle = np.where(axis[:, 2] > 0, ang, -ang)
    results = np.concatenate([np.asarray(post["bbox"], np.float64).reshape(n, 4), score.astype(np.float64)[:, None]], 1)
    ddd = np.stack([size[:, 2], size[:, 0], size[:, 1], center[:, 0], center[:, 1], center[:, 2], angle], 1)
    org = np.concatenate([dim.astype(np.float64), loc.astype(np.float64), rot_y[:, None]], 1)
    sub = np.concatenate([sub_t, size, orient], 1)
    depths = loc[:, 2:3].astype(np.float64)
    out = {}
    for name in NUSCENES_TRACKING_NAMES:
        m = np.nonzero(ok & (names == name))[0]
        r = {"results": results[m], "ddd_boxes": ddd[m], "depths": depths[m], "ddd_org_boxes": org[m], "submission": sub[m]}
        if m.size > 0 and nms:
            keep, _ = greedy_nms(r["results"][:, :4], r["results"][:, -1], overlap=0.7 if name in ("bus", "truck") else 0.8, lib=lib)
            k = sorted(set(keep.tolist()))                         # the reference ignores `count`: index 0 is always kept
            r = {key: v[k] for key, v in r.items()}
        out[name] = r
    return out
