"""TEST INFRASTRUCTURE ONLY.  Functional stand-ins for the third-party packages the reference's
tracker imports and that are absent here: `lap` (Jonker-Volgenant assignment),
`cython_bbox` (IoU matrix), `numba` (jit decorator), `sklearn.utils.linear_assignment_` (unused
import).  Both sides of test_reference_tracker.py run through the same shims."""
import sys
import types

import numpy as np
from scipy.optimize import linear_sum_assignment


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
    """lap.lapjv semantics used by utils/matching.py:48: rectangular cost extended to (n+m)^2
    with cost_limit/2 on the extension entries (0 in the bottom-right block); x[i] = column of
    row i or -1, y[j] = row of column j or -1."""
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
