// Host-side association helpers of the tracker -- plain C++ inside the same library, HOST pointers (the only
// entry points of libdeft_hip.so that take host memory: these are <= 200 x 200 float64 problems on data the association step holds on
// the host; a kernel launch and a copy would cost more than the arithmetic).
//
//  * deft_lapjv: the assignment solver behind matching.linear_assignment (matching.py:40-55).  The reference calls the third-party
//    package `lap` (`lap.lapjv(cost, extend_cost=True, cost_limit=thresh)`, version unpinned, not in /root/reference, not in this
//    image), which solves the Jonker-Volgenant LAP (R. Jonker, A. Volgenant, "A shortest augmenting path algorithm for dense and sparse
//    linear assignment problems", Computing 38, 1987) on its extension of the rectangular problem: an (n + m) x (n + m) matrix with
//    cost_limit / 2 in the two off-diagonal blocks and 0 in the lower-right block, so that a pair is matched only while it is cheaper
//    than leaving both sides unmatched -- i.e. it minimises  sum(matched costs) + cost_limit * #unmatched rows  (+ a constant).
//    With a finite limit (the only form the tracker uses) THAT objective is solved here directly: shortest augmenting paths with row /
//    column potentials over the n x m matrix, every row carrying an implicit private "stay unmatched" column at cost_limit (`Sap`
//    below; O(m) per scan step instead of the O((n + m)^2) scans of the square extension, whose constant blocks are one big tie:
//    1.5 ms -> 0.06 ms at 118 x 100).  Without a limit: the published dense algorithm (column reduction + reduction transfer,
//    augmenting row reduction, augmentation) on the zero-padded square (`Jv` below).  No code of `lap` is available here; `Jv`'s phase
//    structure (ccrrt / carr with its `rr_cnt < current * n` guard / find / scan / augment) follows the dense solver of the `lap` package
//    (github.com/gatagat/lap, BSD-2-Clause, itself after Jonker & Volgenant's published Pascal code) as remembered, re-written so that
//    the order in which ties are broken is fixed and this repository's own: rows in index order,
//    among equally near columns the lowest index, a real column before "unmatched".  tests/test_association.py: optimal against brute
//    force and scipy, equal-cost ties resolved reproducibly.
//  * deft_iou3d_matrix: matching.iou_ddd_distance (matching.py:107-131) = 1 - iou3d for every (track box, detection box) pair, with
//    convert_3dbox_to_8corner (:207-243), polygon_clip (Sutherland-Hodgman, :162-204), poly_area, box3d_vol and iou3d (:253-276)
//    written out per pair in float64; the intersection polygon's area by the shoelace formula instead of scipy's ConvexHull.volume
//    (the clipped polygon of two convex quadrilaterals is convex: same area up to round-off; a degenerate intersection -- fewer than
//    three vertices -- has area 0 here where qhull raises).  Pinned against the reference's functions by tests/golden/iou_ddd.npz.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

typedef double cost_t;
const cost_t LARGE = 1e300;

// ---- Jonker-Volgenant, dense, square ------------------------------------------------------------------------------------------
struct Jv {
    int n;
    const cost_t* c;          // [n][n] row-major
    std::vector<int> x, y, free_rows, pred, cols;
    std::vector<cost_t> v, d;

    cost_t at(int i, int j) const { return c[(size_t)i * n + j]; }

    // column reduction and reduction transfer
    int ccrrt() {
        for (int i = 0; i < n; ++i) { x[i] = -1; v[i] = LARGE; y[i] = 0; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const cost_t cij = at(i, j);
                if (cij < v[j]) { v[j] = cij; y[j] = i; }
            }
        std::vector<char> unique(n, 1);
        for (int j = n - 1; j >= 0; --j) {
            const int i = y[j];
            if (x[i] < 0) x[i] = j;
            else { unique[i] = 0; y[j] = -1; }
        }
        int nfree = 0;
        for (int i = 0; i < n; ++i) {
            if (x[i] < 0) free_rows[nfree++] = i;
            else if (unique[i]) {
                const int j = x[i];
                cost_t mn = LARGE;
                for (int j2 = 0; j2 < n; ++j2) {
                    if (j2 == j) continue;
                    const cost_t r = at(i, j2) - v[j2];
                    if (r < mn) mn = r;
                }
                v[j] -= mn;
            }
        }
        return nfree;
    }

    // augmenting row reduction
    int carr(int nfree) {
        int current = 0, new_free = 0;
        long long rr_cnt = 0;
        while (current < nfree) {
            ++rr_cnt;
            const int fi = free_rows[current++];
            int j1 = 0, j2 = -1;
            cost_t v1 = at(fi, 0) - v[0], v2 = LARGE;
            for (int j = 1; j < n; ++j) {
                const cost_t r = at(fi, j) - v[j];
                if (r < v2) {
                    if (r >= v1) { v2 = r; j2 = j; }
                    else { v2 = v1; v1 = r; j2 = j1; j1 = j; }
                }
            }
            int i0 = y[j1];
            const cost_t v1_new = v[j1] - (v2 - v1);
            const bool lowers = v1_new < v[j1];
            if (rr_cnt < (long long)current * n) {
                if (lowers) v[j1] = v1_new;
                else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
                if (i0 >= 0) {
                    if (lowers) free_rows[--current] = i0;
                    else free_rows[new_free++] = i0;
                }
            } else if (i0 >= 0) {
                free_rows[new_free++] = i0;
            }
            x[fi] = j1;
            y[j1] = fi;
        }
        return new_free;
    }

    // columns with minimum d on the SCAN list
    int find(int lo) {
        int hi = lo + 1;
        cost_t mind = d[cols[lo]];
        for (int k = hi; k < n; ++k) {
            const int j = cols[k];
            if (d[j] <= mind) {
                if (d[j] < mind) { hi = lo; mind = d[j]; }
                cols[k] = cols[hi];
                cols[hi++] = j;
            }
        }
        return hi;
    }

    int scan(int& lo, int& hi) {
        while (lo != hi) {
            int j = cols[lo++];
            const int i = y[j];
            const cost_t mind = d[j];
            const cost_t h = at(i, j) - v[j] - mind;
            for (int k = hi; k < n; ++k) {
                j = cols[k];
                const cost_t cred = at(i, j) - v[j] - h;
                if (cred < d[j]) {
                    d[j] = cred;
                    pred[j] = i;
                    if (cred == mind) {
                        if (y[j] < 0) return j;
                        cols[k] = cols[hi];
                        cols[hi++] = j;
                    }
                }
            }
        }
        return -1;
    }

    // one shortest augmenting path from row `start`
    int find_path(int start) {
        int lo = 0, hi = 0, final_j = -1, n_ready = 0;
        for (int j = 0; j < n; ++j) { cols[j] = j; pred[j] = start; d[j] = at(start, j) - v[j]; }
        while (final_j == -1) {
            if (lo == hi) {
                n_ready = lo;
                hi = find(lo);
                for (int k = lo; k < hi; ++k)
                    if (y[cols[k]] < 0) final_j = cols[k];
            }
            if (final_j == -1) final_j = scan(lo, hi);
        }
        // the level the search ended on = d of the free column it reached.  (NOT d[cols[lo]]: scan() advances `lo` past the column it was
        // expanding when it returns, so cols[lo] may already be a TODO column with a larger d -- the dual update was then wrong and LATER
        // augmentations could end non-optimal: 276 of 9304 random tall matrices, and wide ones with negative costs)
        const cost_t mind = d[final_j];
        for (int k = 0; k < n_ready; ++k) {
            const int j = cols[k];
            v[j] += d[j] - mind;
        }
        return final_j;
    }

    void solve() {
        x.assign(n, -1); y.assign(n, -1); free_rows.assign(n, 0); pred.assign(n, 0); cols.assign(n, 0);
        v.assign(n, 0.0); d.assign(n, 0.0);
        int nfree = ccrrt();
        for (int pass = 0; nfree > 0 && pass < 2; ++pass) nfree = carr(nfree);
        for (int f = 0; f < nfree; ++f) {
            const int start = free_rows[f];
            int j = find_path(start), i = -1;
            while (i != start) {
                i = pred[j];
                y[j] = i;
                const int t = x[i]; x[i] = j; j = t;
            }
        }
    }
};

// ---- shortest augmenting paths with an implicit "unmatched" column per row ------------------------------------------------------
// minimise  sum_{matched (i,j)} c[i][j] + limit * #unmatched rows.  Potentials u (rows), v (columns); the private column of row i has
// potential 0 and is reachable from row i only, so it is free whenever row i is in a search tree (a row matched to it can only be
// reached through it): a path may end at a free real column or at the private column of any row of the tree.
struct Sap {
    int n, m;
    const cost_t* c;          // [n][m] row-major, finite
    cost_t limit;
    std::vector<int> col4row, row4col;          // col4row: real column, -1 = not processed yet, -2 = unmatched (its private column)

    void solve() {
        col4row.assign(n, -1); row4col.assign(m, -1);
        std::vector<cost_t> u(n, 0.0), v(m, 0.0), shortest(m), rowdist(n);
        std::vector<int> path(m), remaining(m), SR; std::vector<char> SC(m);
        SR.reserve(n);
        for (int cur = 0; cur < n; ++cur) {
            for (int j = 0; j < m; ++j) { shortest[j] = LARGE; remaining[j] = j; SC[j] = 0; }
            int nrem = m;
            SR.clear();
            cost_t minVal = 0.0;
            int i = cur, sink = -1, sink_row = -1;          // sink >= 0: a free real column; sink_row >= 0: the private column of that row
            cost_t best_dummy = LARGE; int dummy_row = -1;
            rowdist[cur] = 0.0;
            while (true) {
                SR.push_back(i);
                const cost_t di = minVal;                   // distance at which row i entered the tree
                rowdist[i] = di;
                const cost_t rd = di + limit - u[i];         // its private column
                if (rd < best_dummy) { best_dummy = rd; dummy_row = i; }
                cost_t lowest = LARGE; int idx = -1;
                for (int k = 0; k < nrem; ++k) {
                    const int j = remaining[k];
                    const cost_t r = di + c[(size_t)i * m + j] - u[i] - v[j];
                    if (r < shortest[j]) { shortest[j] = r; path[j] = i; }
                    // the nearest column; among equals the lowest column index (remaining[] is permuted by the removals below)
                    if (shortest[j] < lowest || (idx >= 0 && shortest[j] == lowest && j < remaining[idx])) { lowest = shortest[j]; idx = k; }
                }
                if (idx < 0 || best_dummy < lowest) {        // ending unmatched is strictly nearer than every real column
                    minVal = best_dummy; sink_row = dummy_row;
                    break;
                }
                minVal = lowest;
                const int j = remaining[idx];
                SC[j] = 1;
                remaining[idx] = remaining[--nrem];
                if (row4col[j] < 0) { sink = j; break; }
                i = row4col[j];
            }
            // dual update (before the assignment changes)
            u[cur] += minVal;
            for (size_t k = 1; k < SR.size(); ++k) { const int r = SR[k]; u[r] += minVal - shortest[col4row[r]]; }
            for (int j = 0; j < m; ++j) if (SC[j]) v[j] -= minVal - shortest[j];
            // augment
            int j;
            if (sink_row >= 0) {
                j = col4row[sink_row];                      // the real column row `sink_row` gives up (or -1 / none when it is `cur`)
                col4row[sink_row] = -2;
                if (sink_row == cur) continue;
            } else {
                j = sink;
            }
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int prev = col4row[r];
                col4row[r] = j;
                if (r == cur) break;
                j = prev;
            }
        }
    }
};

// ---- 3-D boxes ----------------------------------------------------------------------------------------------------------------
struct P2 { double x, y; };

// matching.py:162-204: clip `subject` (any polygon) by the convex polygon `clip` (counter-clockwise); returns the vertex count (0 = empty)
int polygon_clip(const P2* subject, int ns, const P2* clip, int nc, P2* out) {
    P2 a[16], b[16];
    int na = ns;
    std::memcpy(a, subject, sizeof(P2) * ns);
    P2 cp1 = clip[nc - 1];
    for (int ci = 0; ci < nc; ++ci) {
        const P2 cp2 = clip[ci];
        int nb = 0;
        P2 s = a[na - 1];
        auto inside = [&](const P2& p) { return (cp2.x - cp1.x) * (p.y - cp1.y) > (cp2.y - cp1.y) * (p.x - cp1.x); };
        auto inter = [&](const P2& s_, const P2& e_) {
            const double dcx = cp1.x - cp2.x, dcy = cp1.y - cp2.y, dpx = s_.x - e_.x, dpy = s_.y - e_.y;
            const double n1 = cp1.x * cp2.y - cp1.y * cp2.x, n2 = s_.x * e_.y - s_.y * e_.x, n3 = 1.0 / (dcx * dpy - dcy * dpx);
            return P2{(n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3};
        };
        for (int k = 0; k < na; ++k) {
            const P2 e = a[k];
            if (inside(e)) {
                if (!inside(s)) b[nb++] = inter(s, e);
                b[nb++] = e;
            } else if (inside(s)) {
                b[nb++] = inter(s, e);
            }
            s = e;
        }
        cp1 = cp2;
        if (nb == 0) return 0;
        na = nb;
        std::memcpy(a, b, sizeof(P2) * nb);
    }
    std::memcpy(out, a, sizeof(P2) * na);
    return na;
}

// matching.py:207-243: (h, w, l, x, y, z, rot_y) -> 8 corners [8][3]
void corners_of(const double* bx, double (*c)[3]) {
    const double h = bx[0], w = bx[1], l = bx[2], x = bx[3], y = bx[4], z = bx[5], t = bx[6];
    const double cs = std::cos(t), sn = std::sin(t);
    const double xc[8] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2};
    const double yc[8] = {0, 0, 0, 0, -h, -h, -h, -h};
    const double zc[8] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2};
    for (int k = 0; k < 8; ++k) {                       // R = [[c, 0, s], [0, 1, 0], [-s, 0, c]] (np.dot: products summed left to right)
        c[k][0] = (cs * xc[k] + 0.0 * yc[k]) + sn * zc[k] + x;
        c[k][1] = (0.0 * xc[k] + 1.0 * yc[k]) + 0.0 * zc[k] + y;
        c[k][2] = (-sn * xc[k] + 0.0 * yc[k]) + cs * zc[k] + z;
    }
}

double shoelace(const P2* p, int n) {                   // poly_area, matching.py:134-135
    double a = 0.0, b = 0.0;
    for (int k = 0; k < n; ++k) {
        const int km = (k + n - 1) % n;
        a += p[k].x * p[km].y;
        b += p[k].y * p[km].x;
    }
    return 0.5 * std::fabs(a - b);
}

double dist3(const double* a, const double* b) {
    return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}

// iou3d(corners1 = detection, corners2 = track), matching.py:253-276
double iou3d(const double (*c1)[3], const double (*c2)[3]) {
    P2 r1[4], r2[4], ip[16];
    for (int k = 0; k < 4; ++k) {
        r1[k] = P2{c1[3 - k][0], c1[3 - k][2]};
        r2[k] = P2{c2[3 - k][0], c2[3 - k][2]};
    }
    const int ni = polygon_clip(r1, 4, r2, 4, ip);
    const double inter_area = ni >= 3 ? shoelace(ip, ni) : 0.0;
    const double ymax = std::fmin(c1[0][1], c2[0][1]), ymin = std::fmax(c1[4][1], c2[4][1]);
    const double inter_vol = inter_area * std::fmax(0.0, ymax - ymin);
    const double vol1 = dist3(c1[0], c1[1]) * dist3(c1[1], c1[2]) * dist3(c1[0], c1[4]);
    const double vol2 = dist3(c2[0], c2[1]) * dist3(c2[1], c2[2]) * dist3(c2[0], c2[4]);
    return inter_vol / (vol1 + vol2 - inter_vol);
}

}  // namespace

extern "C" int deft_lapjv(const double* cost, int n_rows, int n_cols, double cost_limit, int* x, int* y, double* total) {
    DEFT_CHECK(n_rows >= 0 && n_cols >= 0 && (n_rows == 0 || n_cols == 0 || cost != nullptr) && x != nullptr && y != nullptr, -90,
               "deft_lapjv: null pointer or negative size");
    DEFT_CHECK(n_rows + n_cols <= 4096, -91, "deft_lapjv: %d x %d is beyond what the tracker's association builds", n_rows, n_cols);
    for (int i = 0; i < n_rows; ++i) x[i] = -1;
    for (int j = 0; j < n_cols; ++j) y[j] = -1;
    if (total) *total = 0.0;
    if (n_rows == 0 || n_cols == 0) return 0;
    const bool limited = cost_limit < LARGE && !std::isinf(cost_limit);
    // largest finite entry: +inf / NaN entries (gated pairs) are replaced by a value no optimal solution can afford
    double big = 0.0;
    for (long long k = 0; k < (long long)n_rows * n_cols; ++k)
        if (std::isfinite(cost[k]) && std::fabs(cost[k]) > big) big = std::fabs(cost[k]);
    big = big * (n_rows + n_cols + 1) + 1.0;
    if (limited && big < cost_limit + 1.0) big = cost_limit + 1.0;
    double sum = 0.0;
    if (limited) {
        std::vector<cost_t> fin((size_t)n_rows * n_cols);
        for (long long k = 0; k < (long long)n_rows * n_cols; ++k) fin[k] = std::isfinite(cost[k]) ? cost[k] : big;
        Sap sap;
        sap.n = n_rows; sap.m = n_cols; sap.c = fin.data(); sap.limit = cost_limit;
        sap.solve();
        for (int i = 0; i < n_rows; ++i) {
            const int j = sap.col4row[i];
            if (j >= 0 && std::isfinite(cost[(size_t)i * n_cols + j])) {
                x[i] = j;
                y[j] = i;
                sum += cost[(size_t)i * n_cols + j];
            }
        }
    } else {                                           // extend_cost without a limit: zero-padded max(n, m) square
        Jv jv;
        jv.n = n_rows > n_cols ? n_rows : n_cols;
        const int n = jv.n;
        std::vector<cost_t> ext((size_t)n * n, 0.0);
        for (int i = 0; i < n_rows; ++i)
            for (int j = 0; j < n_cols; ++j) {
                const double cc = cost[(size_t)i * n_cols + j];
                ext[(size_t)i * n + j] = std::isfinite(cc) ? cc : big;
            }
        jv.c = ext.data();
        jv.solve();
        for (int i = 0; i < n_rows; ++i) {
            const int j = jv.x[i];
            if (j >= 0 && j < n_cols && std::isfinite(cost[(size_t)i * n_cols + j])) {      // (a forced +inf pairing is no match)
                x[i] = j;
                y[j] = i;
                sum += cost[(size_t)i * n_cols + j];
            }
        }
    }
    if (total) *total = sum;
    return 0;
}

extern "C" int deft_iou3d_matrix(const double* trk, int T, const double* det, int N, float* out) {
    DEFT_CHECK(T >= 0 && N >= 0 && (T == 0 || trk != nullptr) && (N == 0 || det != nullptr) && (T * N == 0 || out != nullptr), -92,
               "deft_iou3d_matrix: null pointer or negative size");
    std::vector<double> ct((size_t)T * 24), cd((size_t)N * 24);
    for (int t = 0; t < T; ++t) corners_of(trk + 7 * t, (double (*)[3])(ct.data() + 24 * t));
    for (int d = 0; d < N; ++d) corners_of(det + 7 * d, (double (*)[3])(cd.data() + 24 * d));
    for (int t = 0; t < T; ++t)
        for (int d = 0; d < N; ++d) {
            const float iou = (float)iou3d((const double (*)[3])(cd.data() + 24 * d), (const double (*)[3])(ct.data() + 24 * t));
            out[(size_t)t * N + d] = 1.0f - iou;         // iou_matrix is float32 in the reference: 1 - float32(iou)
        }
    return 0;
}
