"""TEST INFRASTRUCTURE ONLY.  Functional stand-ins for the third-party packages the reference's
tracker imports and that are absent here: `lap` (Jonker-Volgenant assignment),
`cython_bbox` (IoU matrix), `numba` (jit decorator), `sklearn.utils.linear_assignment_` (unused
import).  Both sides of test_reference_tracker.py run through the same shims."""
import sys
import types

import numpy as np
from scipy.optimize import linear_sum_assignment


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
    """lap.lapjv semantics used by utils/matching.py:48 (rectangular cost, cost_limit): x[i] = column of row i or -1, y[j] = row of
    column j or -1.  Delegates to deft_amd.association.lapjv so that BOTH sides of the tracker comparisons break exact cost ties the
    same way (the scenes of these tests contain them: a young LSTM track without usable nodes costs exactly the limit, 0.9, against
    every detection -- matched and unmatched are then equally optimal, and which one the real `lap` returns is a property of its JV
    internals that no stand-in can claim).  Optimality itself is checked against brute force in tests/test_association.py, and
    `lapjv_square` below (the literal (n+m)^2 extension lap builds) must reach the same objective."""
    from deft_amd.association import lapjv as _lapjv
    return _lapjv(cost, extend_cost=extend_cost, cost_limit=cost_limit, return_cost=return_cost)


def lapjv_square(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
    """The extension lap itself builds: (n+m)^2 with cost_limit/2 on the extension entries (0 in the bottom-right block)."""
    cost = np.asarray(cost, dtype=np.float64)
    n, m = cost.shape
    big = cost_limit / 2.0 if cost_limit < np.inf else cost.max() + 1
    ext = np.full((n + m, n + m), big)
    ext[n:, m:] = 0
    ext[:n, :m] = cost
    r, c = linear_sum_assignment(ext)
    x = np.full(n, -1, dtype=int); y = np.full(m, -1, dtype=int)
    for i, j in zip(r, c):
        if i < n and j < m:
            x[i] = j; y[j] = i
    opt = cost[np.arange(n)[x >= 0], x[x >= 0]].sum()
    return opt, x, y


def bbox_overlaps(boxes, query):
    """cython_bbox.bbox_overlaps (Fast R-CNN): IoU with the +1 pixel convention."""
    b = np.asarray(boxes, dtype=np.float64); q = np.asarray(query, dtype=np.float64)
    out = np.zeros((b.shape[0], q.shape[0]))
    qa = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    for n in range(b.shape[0]):
        iw = np.minimum(b[n, 2], q[:, 2]) - np.maximum(b[n, 0], q[:, 0]) + 1
        ih = np.minimum(b[n, 3], q[:, 3]) - np.maximum(b[n, 1], q[:, 1]) + 1
        ok = (iw > 0) & (ih > 0)
        ua = (b[n, 2] - b[n, 0] + 1) * (b[n, 3] - b[n, 1] + 1) + qa - iw * ih
        out[n] = np.where(ok, iw * ih / ua, 0.0)
    return out


def install():
    def mod(name, **kw):
        m = types.ModuleType(name)
        for k, v in kw.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("lap", lapjv=lapjv)
    mod("cython_bbox", bbox_overlaps=bbox_overlaps)
    mod("numba", jit=lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f)))
    sk = sys.modules.get("sklearn.utils")
    if sk is None:
        import sklearn.utils as sk  # noqa: F401
    mod("sklearn.utils.linear_assignment_", linear_assignment=None)
    if not hasattr(np, "float"):
        np.float = float      # tracker.py:163, 886; matching.py:67-73 use the alias removed in numpy 1.24


class _Anything:
    """Import-time stand-in for classes the reference's detector.py names but the 2-D path never touches."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def get_affine_transform_3pt(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix M with M @ [x, y, 1] = (x', y') for three point pairs (float64 solve)."""
    src = np.asarray(src, dtype=np.float64); dst = np.asarray(dst, dtype=np.float64)
    A = np.concatenate([src, np.ones((3, 1))], 1)
    return np.linalg.solve(A, dst).T


class Quaternion:
    """Functional stand-in for `pyquaternion.Quaternion` (absent from this image: PARITY UNPINNED), covering what detector.py:236-276
    uses: the (axis=, angle=) and 4-sequence constructors, w / x / y / z, the Hamilton product, `rotation_matrix`, and `angle` / `axis`
    as pyquaternion defines them (angle = 2 atan2(|v|, w) wrapped to (-pi, pi]; axis = v / |v|, zeros when |v| < 1e-17).  Written from
    pyquaternion's documented behaviour; the rotation matrix goes through scipy, independent of deft_amd.postprocess' quaternion code."""

    def __init__(self, *args, axis=None, angle=None, **kw):
        if axis is not None:
            a = np.asarray(axis, np.float64)
            a = a / np.linalg.norm(a)
            self.q = np.r_[np.cos(angle / 2.0), a * np.sin(angle / 2.0)].astype(np.float64)
        elif len(args) == 1 and isinstance(args[0], Quaternion):
            self.q = args[0].q.copy()
        elif len(args) == 1:
            self.q = np.asarray(args[0], np.float64).reshape(4).copy()
        elif len(args) == 4:
            self.q = np.asarray(args, np.float64)
        else:
            raise TypeError("Quaternion stub: unsupported constructor %r %r" % (args, kw))

    w = property(lambda self: self.q[0]); x = property(lambda self: self.q[1])
    y = property(lambda self: self.q[2]); z = property(lambda self: self.q[3])

    def __mul__(self, o):
        a1, b1, c1, d1 = self.q
        a2, b2, c2, d2 = o.q
        return Quaternion([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                           a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2])

    def _unit(self):
        n = np.linalg.norm(self.q)
        return self.q / n if n > 0 else self.q

    @property
    def rotation_matrix(self):
        from scipy.spatial.transform import Rotation
        u = self._unit()
        return Rotation.from_quat([u[1], u[2], u[3], u[0]]).as_matrix()

    @property
    def angle(self):
        u = self._unit()
        th = 2.0 * np.arctan2(np.linalg.norm(u[1:]), u[0])
        r = ((th + np.pi) % (2 * np.pi)) - np.pi
        return np.pi if r == -np.pi else r

    @property
    def axis(self):
        u = self._unit()
        n = np.linalg.norm(u[1:])
        return np.zeros(3) if n < 1e-17 else u[1:] / n


class Box:
    """Functional stand-in for `nuscenes.utils.data_classes.Box` (absent: PARITY UNPINNED): center / wlh / orientation, translate()
    and rotate() as the devkit defines them (center kept in the dtype it was given, so the first translate of detector.py:258 is
    float32 arithmetic; rotate() multiplies on the left)."""

    def __init__(self, center, size, orientation, label=np.nan, score=np.nan, velocity=(np.nan, np.nan, np.nan), name=None, token=None):
        self.center, self.wlh, self.orientation = np.array(center), np.array(size), orientation
        self.name, self.token = name, token

    def translate(self, x):
        self.center += x

    def rotate(self, quaternion):
        self.center = np.dot(quaternion.rotation_matrix, self.center)
        self.orientation = quaternion * self.orientation


def install_detector_stubs():
    """What `src/lib/detector.py` imports at module level beyond the tracker's needs: progress,
    nuscenes.eval.*, pycocotools (via dataset_factory) as names only; functional `pyquaternion.Quaternion` and nuscenes `Box` (the
    classes above: the nuScenes branch of Detector.run calls them), and a functional cv2.getAffineTransform (utils/image.py:get_affine_transform; post-processing of the 2-D datasets)."""
    def mod(name, **kw):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in kw.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("progress"); mod("progress.bar", Bar=_Anything)
    mod("pyquaternion", Quaternion=Quaternion)
    mod("nuscenes", NuScenes=_Anything)
    for n, attrs in (("nuscenes.utils", {}), ("nuscenes.utils.data_classes", {"Box": Box}), ("nuscenes.eval", {}),
                     ("nuscenes.eval.common", {}), ("nuscenes.eval.common.data_classes", {"EvalBoxes": _Anything}),
                     ("nuscenes.eval.common.config", {}), ("nuscenes.eval.tracking", {}),
                     ("nuscenes.eval.tracking.data_classes", {"TrackingBox": _Anything}), ("nuscenes.eval.tracking.evaluate", {}),
                     ("nuscenes.eval.detection", {}), ("nuscenes.eval.detection.data_classes", {"DetectionBox": _Anything}),
                     ("nuscenes.eval.detection.evaluate", {})):
        mod(n, **attrs)
    coco = mod("pycocotools.coco", COCO=_Anything)
    mod("pycocotools.cocoeval", COCOeval=_Anything)
    mod("pycocotools", coco=coco)
    mod("cv2", getAffineTransform=get_affine_transform_3pt)
