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


def install_detector_stubs():
    """What `src/lib/detector.py` imports at module level beyond the tracker's needs: progress,
    pyquaternion, nuscenes.*, pycocotools (via dataset_factory) as names only, and a functional
    cv2.getAffineTransform (utils/image.py:get_affine_transform; post-processing of the 2-D datasets)."""
    def mod(name, **kw):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in kw.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("progress"); mod("progress.bar", Bar=_Anything)
    mod("pyquaternion", Quaternion=_Anything)
    mod("nuscenes", NuScenes=_Anything)
    for n, attrs in (("nuscenes.utils", {}), ("nuscenes.utils.data_classes", {"Box": _Anything}), ("nuscenes.eval", {}),
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
