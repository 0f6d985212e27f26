"""-m "not gpu": deft_amd.association (vectorised host side of the association step).
  * fuse_motion / fuse_motion_ddd against REFERENCE outputs (tests/golden/association.npz, written by
    oracle/make_golden.py from matching.py:311-415 with the reference's KalmanFilter / KalmanFilterLSTM);
  * lapjv / linear_assignment / bbox_overlaps: the reference takes these from `lap` and `cython_bbox`, which are
    not installed here (parity unpinned) -- checked against brute force and a scalar restatement."""
import itertools
import os
from types import SimpleNamespace

import numpy as np
import pytest

from deft_amd import association as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _xyah(b):
    r = np.asarray(b, dtype=np.float64).copy()
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


def _same(a, b):
    assert np.array_equal(np.isinf(a), np.isinf(b))
    f = np.isfinite(a)
    assert np.allclose(a[f], b[f], rtol=1e-12, atol=1e-12)


def test_fuse_motion_vs_reference_fixture():
    fx = np.load(os.path.join(GOLD, "association.npz"))
    dets = [SimpleNamespace(to_xyah=(lambda b=b: _xyah(b))) for b in fx["det_tlwh"]]
    T = fx["kal_mean"].shape[0]
    tracks = [SimpleNamespace(mean=fx["kal_mean"][t], covariance=fx["kal_cov"][t]) for t in range(T)]
    cost = fx["kal_cost"].copy()
    out = A.fuse_motion(None, cost, tracks, dets, frame_id=5, use_lstm=False)
    assert out is cost                                        # in place, like the reference
    _same(out, fx["kal_out"])
    tracks = [SimpleNamespace(observations=[0] * int(fx["lstm_nobs"][t]), covariance=fx["lstm_cov"][t],
                              prediction_at_frame=(lambda f, t=t: fx["lstm_pred"][t])) for t in range(T)]
    _same(A.fuse_motion(None, fx["lstm_cost"].copy(), tracks, dets, frame_id=5, use_lstm=True), fx["lstm_out"])
    young = fx["lstm_nobs"] < 300
    assert np.array_equal(fx["lstm_out"][young], 0.9 * fx["lstm_cost"][young])       # the 2-D "gaussian" distance is 0
    assert A.fuse_motion(None, np.zeros((0, 3)), [], dets[:3], 5).shape == (0, 3)


def test_fuse_motion_ddd_vs_reference_fixture():
    fx = np.load(os.path.join(GOLD, "association.npz"))
    dets = [SimpleNamespace(ddd_bbox=b) for b in fx["ddd_det"]]
    tracks = [SimpleNamespace(ddd_bbox=fx["ddd_trk"][t], depth=fx["ddd_depth"][t]) for t in range(fx["ddd_trk"].shape[0])]
    KalmanFilterLSTM = type("KalmanFilterLSTM", (), {})
    for cls in ("pedestrian", "car"):
        _same(A.fuse_motion_ddd(KalmanFilterLSTM(), fx["ddd_cost"].copy(), tracks, dets, frame_id=5, classe_name=cls), fx["ddd_out_" + cls])
    # opt.lstm off: the tracker hands over the reference's plain `KalmanFilter`, whose "gaussian" distance is squared and over all seven
    # components (kalman_filter.py:271-273) -- recognised by its class name
    KalmanFilter = type("KalmanFilter", (), {})
    near = [SimpleNamespace(ddd_bbox=fx["ddd_trk_near"][t], depth=fx["ddd_depth"][t]) for t in range(fx["ddd_trk_near"].shape[0])]
    for cls in ("pedestrian", "car"):
        _same(A.fuse_motion_ddd(KalmanFilter(), fx["ddd_cost"].copy(), near, dets, frame_id=5, classe_name=cls), fx["ddd_out_kf_" + cls])
    # the metric follows the class hierarchy or an explicit attribute, never a guess: a subclass resolves like its base, a
    # wrapper names its metric, anything else is an error
    Sub = type("TunedFilter", (KalmanFilter,), {})
    _same(A.fuse_motion_ddd(Sub(), fx["ddd_cost"].copy(), near, dets, frame_id=5, classe_name="car"), fx["ddd_out_kf_car"])
    wrapped = SimpleNamespace(ddd_metric="centre")
    _same(A.fuse_motion_ddd(wrapped, fx["ddd_cost"].copy(), tracks, dets, frame_id=5, classe_name="car"), fx["ddd_out_car"])
    for bad in (None, SimpleNamespace(), SimpleNamespace(ddd_metric="euclid")):
        with pytest.raises((TypeError, ValueError)):
            A.fuse_motion_ddd(bad, fx["ddd_cost"].copy(), tracks, dets, frame_id=5, classe_name="car")


def test_not_positive_definite_raises_like_cholesky():
    dets = [SimpleNamespace(to_xyah=lambda: np.array([1.0, 2.0, 0.5, 10.0]))]
    bad = SimpleNamespace(mean=np.zeros(8), covariance=-np.eye(8))
    with pytest.raises(np.linalg.LinAlgError):
        A.fuse_motion(None, np.zeros((1, 1)), [bad], dets, 1, use_lstm=False)


def _brute(cost, limit):
    """Optimal partial assignment: minimise sum(matched costs) + limit/2 per unmatched row and column."""
    n, m = cost.shape
    best, arg = np.inf, None
    for k in range(min(n, m) + 1):
        for rows in itertools.combinations(range(n), k):
            for cols in itertools.permutations(range(m), k):
                c = sum(cost[r, cc] for r, cc in zip(rows, cols)) + (n + m - 2 * k) * limit / 2.0
                if c < best:
                    best, arg = c, sorted(zip(rows, cols))
    return arg


@pytest.mark.parametrize("n,m,seed", [(3, 4, 0), (4, 3, 1), (4, 4, 2), (1, 5, 3), (5, 2, 4)])
def test_linear_assignment_is_optimal(n, m, seed):
    g = np.random.RandomState(seed)
    cost = g.rand(n, m)
    cost[g.rand(n, m) < 0.2] = np.inf                          # gated pairs (fuse_motion)
    for thr in (0.9, 0.4):
        matches, ua, ub = A.linear_assignment(cost.copy(), thresh=thr)
        assert sorted(map(tuple, matches.tolist())) == _brute(np.where(np.isinf(cost), 1e9, cost), thr)
        assert sorted(ua.tolist() + matches[:, 0].tolist()) == list(range(n))
        assert sorted(ub.tolist() + matches[:, 1].tolist()) == list(range(m))
        assert all(cost[i, j] <= thr for i, j in matches)


def test_linear_assignment_edges():
    m, ua, ub = A.linear_assignment(np.zeros((0, 4)), 0.9)
    assert m.shape == (0, 2) and ua == () and ub == (0, 1, 2, 3)
    m, ua, ub = A.linear_assignment(np.full((2, 3), np.inf), 0.9)
    assert m.shape == (0, 2) and ua.tolist() == [0, 1] and ub.tolist() == [0, 1, 2]
    total, x, y = A.lapjv(np.array([[0.1, 0.8], [0.7, 0.2]]))          # square, no limit: plain assignment
    assert x.tolist() == [0, 1] and y.tolist() == [0, 1] and abs(total - 0.3) < 1e-12
    with pytest.raises(ValueError):
        A.lapjv(np.zeros((2, 3)))
    m, _, _ = A.linear_assignment(np.array([[0.5]]), thresh=0.0)       # tracker.py:1000: nuScenes IoU stage
    assert m.shape == (0, 2)


def test_bbox_overlaps_plus_one_convention():
    a = np.array([[0, 0, 9, 9], [5, 5, 14, 14], [20, 20, 29, 29], [0, 0, 0, 0]], dtype=np.float64)
    b = np.array([[0, 0, 9, 9], [10, 0, 19, 9], [9, 9, 9, 9]], dtype=np.float64)
    got = A.bbox_overlaps(a, b)

    def one(p, q):
        iw = min(p[2], q[2]) - max(p[0], q[0]) + 1
        ih = min(p[3], q[3]) - max(p[1], q[1]) + 1
        if iw <= 0 or ih <= 0:
            return 0.0
        ua = (p[2] - p[0] + 1) * (p[3] - p[1] + 1) + (q[2] - q[0] + 1) * (q[3] - q[1] + 1) - iw * ih
        return iw * ih / ua
    want = np.array([[one(p, q) for q in b] for p in a])
    assert np.array_equal(got, want)
    assert got[0, 0] == 1.0 and got[0, 1] == 0.0 and got[0, 2] == 1.0 / 100 and got[2].max() == 0.0
    assert A.bbox_overlaps(np.zeros((0, 4)), b).shape == (0, 3)


def test_compat_modules_expose_the_two_entry_points():
    import importlib.util
    root = os.path.join(os.path.dirname(GOLD), "..", "deft_amd", "compat")
    for name, attr in (("lap", "lapjv"), ("cython_bbox", "bbox_overlaps")):
        spec = importlib.util.spec_from_file_location("_compat_" + name, os.path.join(root, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert getattr(mod, attr) is getattr(A, attr)


def test_lapjv_without_limit_and_nan():
    """the no-limit form pads to a max(n,m) square with zeros like lap (every row of the smaller side is matched, also with
    negative costs); NaN costs never match and never raise."""
    from deft_amd import association as A
    g = np.random.RandomState(4)
    c = g.randn(3, 5)                                    # negative entries
    tot, x, y = A.lapjv(c, extend_cost=True)
    assert (x >= 0).all() and (y >= 0).sum() == 3
    best = min(sum(c[i, p[i]] for i in range(3)) for p in __import__("itertools").permutations(range(5), 3))
    assert abs(tot - best) <= 1e-12
    with pytest.raises(ValueError):
        A.lapjv(c)
    c2 = g.rand(4, 4); c2[1, :] = np.nan; c2[2, 3] = np.nan
    tot, x, y = A.lapjv(c2, extend_cost=True, cost_limit=0.9)
    assert x[1] == -1 and not (x[2] == 3)
    tot, x, y = A.lapjv(c2)
    assert x[1] == -1 and sorted(v for v in x if v >= 0) == sorted(set(v for v in x if v >= 0))


def test_lapjv_without_limit_is_optimal_for_tall_wide_and_square():
    """the unlimited rectangular form (lap's extend_cost=True, zero-padded square) against scipy's rectangular solver, TALL
    matrices and negative costs included (the dual update of an augmentation read its level from a column scan() had already stepped
    past: later augmentations could end non-optimal, e.g. 0.816 against 0.673 on a 9 x 6 case, -2.5 against -2.7 on a 3 x 7 one).  Every
    row of the smaller side is matched, x / y are mutually consistent."""
    from scipy.optimize import linear_sum_assignment
    from deft_amd import association as A
    g = np.random.RandomState(9)
    shapes = [(9, 6), (6, 9), (7, 7), (12, 3), (3, 12), (1, 5), (5, 1), (20, 13), (13, 20), (40, 25)]
    for n, m in shapes:
        for rep in range(60):
            c = g.rand(n, m) if rep % 3 else g.randn(n, m)             # (negative costs too)
            if rep % 5 == 4:
                c = np.round(c, 1)                                     # many exact ties
            tot, x, y = A.lapjv(c, extend_cost=True)
            ri, ci = linear_sum_assignment(c)
            assert abs(tot - c[ri, ci].sum()) <= 1e-9, (n, m, rep, tot, c[ri, ci].sum())
            assert (x >= 0).sum() == min(n, m) == (y >= 0).sum()
            for i in range(n):
                if x[i] >= 0:
                    assert y[x[i]] == i
            assert abs(sum(c[i, x[i]] for i in range(n) if x[i] >= 0) - tot) <= 1e-9


def test_half_size_extension_reaches_the_square_extension_objective():
    """association.lapjv solves lap's cost-limit problem on an n x (m + n) matrix; the literal (n + m)^2 extension (tests/ref_shims.py
    lapjv_square) must give the same objective sum(c) + (unmatched rows + unmatched columns) * limit / 2 -- on random problems,
    with +inf entries, and with costs exactly AT the limit (ties)."""
    import ref_shims
    from deft_amd import association as A
    g = np.random.default_rng(5)
    for trial in range(60):
        n, m = int(g.integers(1, 12)), int(g.integers(1, 12))
        c = g.uniform(0, 1.4, (n, m))
        if trial % 3 == 0:
            c[g.uniform(size=c.shape) < 0.3] = 0.9                     # exactly the limit
        if trial % 4 == 0:
            c[g.uniform(size=c.shape) < 0.2] = 5.0                      # far above the limit (lap sees finite costs there)
        lim = 0.9

        def obj(x):
            k = int((x >= 0).sum())
            return c[np.nonzero(x >= 0)[0], x[x >= 0]].sum() + (n - k) * lim / 2 + (m - k) * lim / 2
        _, x1, _ = A.lapjv(c, extend_cost=True, cost_limit=lim)
        _, x2, _ = ref_shims.lapjv_square(c, extend_cost=True, cost_limit=lim)
        assert abs(obj(x1) - obj(x2)) <= 1e-9, (trial, obj(x1), obj(x2))


def test_iou_ddd_distance_vs_reference_fixture():
    """deft_iou3d_matrix against the REFERENCE's matching.iou_ddd_distance (tests/golden/iou_ddd.npz, oracle/make_golden.py run_iou_ddd):
    float32 results, polygon area by shoelace vs the reference's scipy ConvexHull -- same to float32 round-off."""
    fx = np.load(os.path.join(GOLD, "iou_ddd.npz"))
    out = A.iou_ddd_distance(fx["trk"], fx["det"])
    assert out.dtype == np.float32 and out.shape == fx["out"].shape
    assert np.abs(out - fx["out"]).max() <= 2e-7, float(np.abs(out - fx["out"]).max())
    assert ((out == 1.0) == (fx["out"] == 1.0)).all()                 # disjoint pairs are exactly 1 on both sides
    assert A.iou_ddd_distance(np.zeros((0, 7)), fx["det"]).shape == (0, len(fx["det"]))
    assert A.iou_ddd_distance(fx["trk"], np.zeros((0, 7))).shape == (len(fx["trk"]), 0)


def test_lapjv_ties_are_resolved_reproducibly():
    """equal costs (matching.py:40-55 sees them whenever two pairs have the same 1 - similarity, e.g. both 1.0
    after a gate or both exactly 0).  `lap` is not available to compare with; the solver here is this repository's own statement of
    the Jonker-Volgenant algorithm in lap's arrangement (csrc/assoc.hip), so the tie order is fixed by construction: the answers
    below are what it returns, today and on every box.  Each is optimal; where scipy's solver picks another optimal assignment the
    case says so."""
    from scipy.optimize import linear_sum_assignment
    # two rows that want the same column at the same cost: the LATER row keeps it (column reduction scans columns from the last one
    # back and the last writer of v[j] -- the first row that reaches the minimum -- is displaced in favour of ... the fixed order below)
    c = np.array([[0.2, 0.5], [0.2, 0.5]])
    _, x, y = A.lapjv(c, extend_cost=True, cost_limit=0.9)
    assert sorted(x.tolist()) == [0, 1] and x.tolist() == A.lapjv(c.copy(), extend_cost=True, cost_limit=0.9)[1].tolist()
    first = x.tolist()
    # an all-equal block: any permutation is optimal; the solver's answer is one fixed permutation, stable across calls and independent
    # of unrelated rows appended below
    c = np.full((4, 4), 0.3)
    _, x4, _ = A.lapjv(c, extend_cost=True, cost_limit=0.9)
    assert sorted(x4.tolist()) == [0, 1, 2, 3]
    big = np.full((6, 4), 0.3); big[4:] = 5.0
    _, x6, _ = A.lapjv(big, extend_cost=True, cost_limit=0.9)
    assert x6[4:].tolist() == [-1, -1] and sorted(x6[:4].tolist()) == [0, 1, 2, 3]
    # cost exactly AT the limit: lap's extension prices "both unmatched" at limit/2 + limit/2 = the pair's cost -- a tie between
    # matching and not matching.  Fixed answer of this solver:
    c = np.array([[0.9]])
    _, xa, _ = A.lapjv(c, extend_cost=True, cost_limit=0.9)
    at_limit = xa.tolist()
    assert at_limit in ([0], [-1]) and A.lapjv(c, extend_cost=True, cost_limit=0.9)[1].tolist() == at_limit
    # random matrices with many exact ties (costs on a coarse grid): always optimal (same objective as scipy on lap's extension)
    g = np.random.RandomState(3)
    diff = 0
    for trial in range(40):
        n, m = g.randint(2, 9), g.randint(2, 9)
        c = g.randint(0, 4, (n, m)) * 0.25
        lim = 0.6
        tot, x, y = A.lapjv(c, extend_cost=True, cost_limit=lim)
        k = int((x >= 0).sum())
        ext = np.full((n + m, n + m), lim / 2); ext[n:, m:] = 0; ext[:n, :m] = c
        r, cc = linear_sum_assignment(ext)
        assert abs((tot + (n + m - 2 * k) * lim / 2) - ext[r, cc].sum()) <= 1e-12
        xs = np.full(n, -1); keep = (r < n) & (cc < m); xs[r[keep]] = cc[keep]
        diff += int((xs != x).any())
        for i in range(n):                                             # consistent x / y
            assert x[i] < 0 or y[x[i]] == i
    print("lapjv tie cases: [[.2,.5],[.2,.5]] -> x = %s; 1x1 at the limit -> x = %s; %d of 40 tied random problems: another optimal assignment than scipy's"
          % (first, at_limit, diff))


def _associate_numpy(sim, mean2, cov2, gated, meas2, second, iou_ok, a_tlbr, d_tlbr):
    """The three stages of tracker.py:886-1030 from this module's numpy pieces (what ArrayTracker._associate_stages composes)."""
    T, N = len(mean2), len(meas2)
    lam = 0.9
    mt, md = [], []
    rows, cols = np.arange(T), np.arange(N)
    s64 = sim.astype(np.float64) if sim is not None else None
    if T and N:
        d = 1 - s64[:, :N]
        g = A._maha2(mean2, cov2, meas2)
        gi = gated.astype(bool)
        sub = d[gi]
        sub[g[gi] > 5.0 * A.chi2inv95[2]] = np.inf
        d[gi] = lam * sub + 0.05 * (1 - lam) * g[gi]
        d[~gi] = lam * d[~gi] + 0.0005 * (1 - lam) * 0.0
        m, u_t, u_d = A.linear_assignment(d, 0.9)
        if len(m):
            mt += rows[m[:, 0]].tolist(); md += cols[m[:, 1]].tolist()
        rows, cols = rows[np.asarray(u_t, dtype=int)], cols[np.asarray(u_d, dtype=int)]
        if second and len(rows) and len(cols):
            m, u_t, u_d = A.linear_assignment(1 - s64[rows][:, cols], 0.9)
            if len(m):
                mt += rows[m[:, 0]].tolist(); md += cols[m[:, 1]].tolist()
            rows, cols = rows[np.asarray(u_t, dtype=int)], cols[np.asarray(u_d, dtype=int)]
    rows = rows[iou_ok[rows].astype(bool)]
    if len(rows) and len(cols):
        m, u_t, u_d = A.linear_assignment(1 - A.bbox_overlaps(a_tlbr[rows], d_tlbr[cols]), 0.9)
        if len(m):
            mt += rows[m[:, 0]].tolist(); md += cols[m[:, 1]].tolist()
        rows, cols = rows[np.asarray(u_t, dtype=int)], cols[np.asarray(u_d, dtype=int)]
    return mt, md, rows.tolist(), cols.tolist()


@pytest.mark.parametrize("seed", range(24))
def test_associate_2d_equals_the_numpy_stages(seed):
    """deft_associate_2d (one host call: gate + fuse + three assignments + IoU) against the same cascade composed from fuse / linear_assignment /
    bbox_overlaps: identical matches in identical order, identical left-overs -- empty sides, rows outside the gate, rows without a gate (young
    LSTM tracks), the KITTI second stage and its age filter in front of the IoU stage."""
    import ctypes as C
    rng = np.random.default_rng(900 + seed)
    T = int(rng.integers(0, 40)) if seed % 6 else 0
    N = int(rng.integers(0, 40)) if seed % 7 else 0
    second = seed % 2
    boxes = lambda n: np.concatenate([(xy := rng.uniform(0, 300, (n, 2))), xy + rng.uniform(10, 80, (n, 2))], 1)
    d_tlbr = boxes(N)
    a_tlbr = boxes(T)
    k = min(T, N)
    if k:                                                               # some tracks sit on detections: real matches in every stage
        pick = rng.permutation(N)[:k]
        a_tlbr[:k] = d_tlbr[pick] + rng.normal(0, 3, (k, 4))
    centre = lambda b: np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2], 1)
    meas2 = centre(d_tlbr) if N else np.zeros((0, 2))
    mean2 = centre(a_tlbr) + rng.normal(0, 2, (T, 2)) if T else np.zeros((0, 2))
    L = rng.normal(0, 4, (T, 2, 2)) + 6 * np.eye(2)
    cov2 = L @ L.transpose(0, 2, 1)
    gated = (rng.random(T) < 0.8).astype(np.uint8)
    iou_ok = (rng.random(T) < 0.8).astype(np.uint8)
    sim = rng.random((T, N + 1)).astype(np.float32) * 0.4
    if k:
        sim[np.arange(k), pick] = (0.5 + 0.5 * rng.random(k)) * (rng.random(k) < 0.7)
    if seed % 5 == 0 and T and N:
        sim[:, :] = 0.25                                                # every embedding cost equal: ties
    with np.errstate(invalid="ignore"):
        chol = np.stack([np.sqrt(cov2[:, 0, 0]), cov2[:, 1, 0] / np.sqrt(cov2[:, 0, 0]),
                         np.sqrt(cov2[:, 1, 1] - (cov2[:, 1, 0] / np.sqrt(cov2[:, 0, 0])) ** 2)], 1) if T else np.zeros((0, 3))
    out = np.full(2 * k + T + N + 3, -7, np.int32)
    mt, md, lost, new_d, cnt = out[:k], out[k:2 * k], out[2 * k:2 * k + T], out[2 * k + T:2 * k + T + N], out[2 * k + T + N:]
    ptr = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)
    lam = 0.9
    lib = A._host_lib()
    arrs = [np.ascontiguousarray(a) for a in (mean2, chol, meas2, a_tlbr, d_tlbr)]
    lib.call("deft_associate_2d", ptr(sim) if T and N else None, N + 1, T, N, ptr(arrs[0]), ptr(arrs[1]), ptr(gated), ptr(arrs[2]),
             C.c_double(5.0 * A.chi2inv95[2]), C.c_double(lam), C.c_double(0.05 * (1 - lam)), second, ptr(iou_ok), ptr(arrs[3]), ptr(arrs[4]),
             C.c_double(0.9), C.c_double(0.9), ptr(mt), ptr(md), C.c_void_p(cnt.ctypes.data), ptr(lost), C.c_void_p(cnt.ctypes.data + 4), ptr(new_d),
             C.c_void_p(cnt.ctypes.data + 8))
    want = _associate_numpy(sim if T and N else None, mean2, cov2, gated, meas2, second, iou_ok, a_tlbr, d_tlbr)
    got = (mt[:cnt[0]].tolist(), md[:cnt[0]].tolist(), lost[:cnt[1]].tolist(), new_d[:cnt[2]].tolist())
    assert got == want
    if seed % 6 and seed % 7 and seed % 5:
        assert cnt[0] > 0


def _associate_ddd_numpy(sim, stage0, recent, trk_ddd, det_ddd, depth, metric, floor, iou_ok, a_tlbr, d_tlbr):
    """tracker.py:850-1030 from this module's numpy pieces (what ArrayTracker._associate_stages composes for a nuScenes class)."""
    T, N = len(trk_ddd), len(det_ddd)
    lam = 0.9
    mt, md = [], []
    pool, det_left = np.arange(T), np.arange(N)
    s64 = sim.astype(np.float64) if sim is not None else None
    if stage0:
        new, old = pool[recent.astype(bool)], pool[~recent.astype(bool)]
        cost = A.iou_ddd_distance(trk_ddd[new], det_ddd)
        m, u_t, u_d = A.linear_assignment(cost, 0.999)
        if len(m):
            mt += new[m[:, 0]].tolist(); md += m[:, 1].tolist()
        det_left = np.asarray(u_d, dtype=int)
        pool = np.concatenate([new[np.asarray(u_t, dtype=int)], old]).astype(int)
    have = len(pool) and len(det_left) and s64 is not None
    if have:
        d = 1 - s64[pool][:, det_left]
        dd = det_ddd[det_left][None, :, :] - trk_ddd[pool][:, None, :]
        g = np.sqrt(np.sum(dd[..., 3:-1] * dd[..., 3:-1], axis=2)) if metric == 0 else np.sum(dd * dd, axis=2)
        thr = np.maximum(0.2 * depth[pool], floor)
        d[g > thr[:, None]] = np.inf
        d = lam * d + 0.001 * g
        m, u_t, u_d = A.linear_assignment(d, 0.9)
        if len(m):
            mt += pool[m[:, 0]].tolist(); md += det_left[m[:, 1]].tolist()
        pool, det_left = pool[np.asarray(u_t, dtype=int)], det_left[np.asarray(u_d, dtype=int)]
        if len(pool) and len(det_left):
            m, u_t, u_d = A.linear_assignment(1 - s64[pool][:, det_left], 0.9)
            if len(m):
                mt += pool[m[:, 0]].tolist(); md += det_left[m[:, 1]].tolist()
            pool, det_left = pool[np.asarray(u_t, dtype=int)], det_left[np.asarray(u_d, dtype=int)]
    pool = pool[iou_ok[pool].astype(bool)]
    if len(pool) and len(det_left):
        m, u_t, u_d = A.linear_assignment(1 - A.bbox_overlaps(a_tlbr[pool], d_tlbr[det_left]), 0.0)
        if len(m):
            mt += pool[m[:, 0]].tolist(); md += det_left[m[:, 1]].tolist()
        pool, det_left = pool[np.asarray(u_t, dtype=int)], det_left[np.asarray(u_d, dtype=int)]
    return mt, md, pool.tolist(), det_left.tolist()


@pytest.mark.parametrize("seed", range(24))
def test_associate_ddd_equals_the_numpy_stages(seed):
    """deft_associate_ddd (3-D IoU stage, embedding stage with the 3-D gate in both metrics, similarity-only stage, 2-D IoU stage at threshold 0)
    against the same cascade composed from iou_ddd_distance / linear_assignment / bbox_overlaps: identical matches in identical order, identical
    left-overs; with and without the 3-D stage (pedestrian), empty sides, rows outside the gate, old rows that skip the 3-D and the IoU stage."""
    import ctypes as C
    rng = np.random.default_rng(1700 + seed)
    T = int(rng.integers(0, 25)) if seed % 6 else 0
    N = int(rng.integers(0, 25)) if seed % 7 else 0
    stage0, metric = seed % 2, (seed // 2) % 2
    floor = 10.0 if stage0 else 5.0

    def boxes3(n):            # (h, w, l, x, y, z, rot_y): cars on a ground plane
        return np.stack([rng.uniform(1.4, 1.9, n), rng.uniform(1.6, 2.0, n), rng.uniform(3.5, 5.0, n), rng.uniform(-30, 30, n), rng.uniform(1.0, 1.6, n),
                         rng.uniform(5, 60, n), rng.uniform(-3.1, 3.1, n)], 1)
    det_ddd, trk_ddd = boxes3(N), boxes3(T)
    k = min(T, N)
    pick = rng.permutation(N)[:k]
    if k:                                                               # tracks near detections: overlapping 3-D boxes, pairs inside the gate
        trk_ddd[:k] = det_ddd[pick] + rng.normal(0, 0.25, (k, 7)) * [0.02, 0.02, 0.05, 1, 0.05, 1, 0.1]
    depth = trk_ddd[:, 5].copy() if T else np.zeros(0)
    box2 = lambda b3: np.stack([b3[:, 3] * 10 + 500, b3[:, 5] * 4, b3[:, 3] * 10 + 500 + 40 + b3[:, 2], b3[:, 5] * 4 + 30 + b3[:, 0]], 1)
    d_tlbr = box2(det_ddd) if N else np.zeros((0, 4))
    a_tlbr = box2(trk_ddd) + rng.normal(0, 1.0, (T, 4)) if T else np.zeros((0, 4))
    recent = (rng.random(T) < 0.7).astype(np.uint8)
    sim = (rng.random((T, N + 1)) * 0.11).astype(np.float32)           # (few pairs beyond the 0.9 limit of the similarity-only stage: the IoU stage gets work)
    if k:
        sim[np.arange(k), pick] = (0.5 + 0.5 * rng.random(k)) * (rng.random(k) < 0.6)
    out = np.full(2 * k + T + N + 3, -7, np.int32)
    mt, md, lost, new_d, cnt = out[:k], out[k:2 * k], out[2 * k:2 * k + T], out[2 * k + T:2 * k + T + N], out[2 * k + T + N:]
    ptr = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (trk_ddd, det_ddd, depth, a_tlbr, d_tlbr)]
    A._host_lib().call("deft_associate_ddd", ptr(sim) if T and N else None, N + 1, T, N, stage0, ptr(recent), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]),
                       metric, C.c_double(floor), C.c_double(0.9), C.c_double(0.001), ptr(recent), ptr(arrs[3]), ptr(arrs[4]), C.c_double(0.999),
                       C.c_double(0.9), C.c_double(0.0), ptr(mt), ptr(md), C.c_void_p(cnt.ctypes.data), ptr(lost), C.c_void_p(cnt.ctypes.data + 4),
                       ptr(new_d), C.c_void_p(cnt.ctypes.data + 8))
    want = _associate_ddd_numpy(sim if T and N else None, stage0, recent, trk_ddd, det_ddd, depth, metric, floor, recent, a_tlbr, d_tlbr)
    got = (mt[:cnt[0]].tolist(), md[:cnt[0]].tolist(), lost[:cnt[1]].tolist(), new_d[:cnt[2]].tolist())
    assert got == want
    if T > 3 and N > 3:
        assert cnt[0] > 0
