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
#include <algorithm>
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

// linear_assignment (matching.py:40-55) with a finite cost_limit on an n x m matrix: x[i] = column of row i or -1, y[j] = row of column j or
// -1.  +inf / NaN entries (gated pairs) are replaced by a value no optimal solution can afford and never reported as matches.
double assign_limited(const double* cost, int n_rows, int n_cols, double cost_limit, int* x, int* y) {
    for (int i = 0; i < n_rows; ++i) x[i] = -1;
    for (int j = 0; j < n_cols; ++j) y[j] = -1;
    if (n_rows == 0 || n_cols == 0) return 0.0;
    const size_t nm = (size_t)n_rows * n_cols;
    double big = 0.0;
    for (size_t k = 0; k < nm; ++k)
        if (std::isfinite(cost[k]) && std::fabs(cost[k]) > big) big = std::fabs(cost[k]);
    big = big * (n_rows + n_cols + 1) + 1.0;
    if (big < cost_limit + 1.0) big = cost_limit + 1.0;
    std::vector<cost_t> fin(nm);
    for (size_t k = 0; k < nm; ++k) fin[k] = std::isfinite(cost[k]) ? cost[k] : big;
    Sap sap;
    sap.n = n_rows; sap.m = n_cols; sap.c = fin.data(); sap.limit = cost_limit;
    sap.solve();
    double sum = 0.0;
    for (int i = 0; i < n_rows; ++i) {
        const int j = sap.col4row[i];
        if (j >= 0 && std::isfinite(cost[(size_t)i * n_cols + j])) {
            x[i] = j;
            y[j] = i;
            sum += cost[(size_t)i * n_cols + j];
        }
    }
    return sum;
}


}  // namespace

extern "C" int deft_lapjv(const double* cost, int n_rows, int n_cols, double cost_limit, int* x, int* y, double* total) {
    DEFT_CHECK(n_rows >= 0 && n_cols >= 0 && (n_rows == 0 || n_cols == 0 || cost != nullptr) && x != nullptr && y != nullptr, -90,
               "deft_lapjv: null pointer or negative size");
    DEFT_CHECK((long long)n_rows * (long long)n_cols < (1ll << 31), -91, "deft_lapjv: %d x %d overflows the 32-bit cost index", n_rows, n_cols);
    for (int i = 0; i < n_rows; ++i) x[i] = -1;
    for (int j = 0; j < n_cols; ++j) y[j] = -1;
    if (total) *total = 0.0;
    if (n_rows == 0 || n_cols == 0) return 0;
    const bool limited = cost_limit < LARGE && !std::isinf(cost_limit);
    double sum = 0.0;
    if (limited) {
        sum = assign_limited(cost, n_rows, n_cols, cost_limit, x, y);
    } else {                                           // extend_cost without a limit: zero-padded max(n, m) square
        // largest finite entry: +inf / NaN entries (gated pairs) are replaced by a value no optimal solution can afford
        double big = 0.0;
        for (long long k = 0; k < (long long)n_rows * n_cols; ++k)
            if (std::isfinite(cost[k]) && std::fabs(cost[k]) > big) big = std::fabs(cost[k]);
        big = big * (n_rows + n_cols + 1) + 1.0;
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

// The association cascade of one frame of Tracker.update on the 2-D datasets (tracker.py:886-1030), host memory, float64 like the reference's numpy:
//   stage 1  embedding distance 1 - similarity, fused with the motion gate (matching.fuse_motion, matching.py:311-371: rows with `gated`:
//            pairs beyond gate_thr on the squared Mahalanobis distance of the detection centre are excluded, w_gate of it is added; the other
//            rows -- LSTM tracks with < 300 observations -- are scaled by lambda only), linear_assignment at thr_embed (matching.py:40-55);
//   stage 2  (second_stage, KITTI: tracker.py:954-980) the same similarities without the motion term on what is left;
//   stage 3  IoU distance (matching.py:71-104, the inclusive-pixel IoU of cython_bbox) between the left-over tracks with `iou_ok` and the
//            left-over detections, linear_assignment at thr_iou.
// Every stage's matches are reported in the order the reference appends them (row index ascending inside a stage).  lost_t: rows of stage 3
// still unmatched; new_d: detections still unmatched.  The arithmetic is the reference's, operation for operation, with no fused multiply-add.
#pragma clang fp contract(off)
extern "C" int deft_associate_2d(const float* sim, int ld, int T, int N, const double* mean2, const double* chol, const unsigned char* gated,
                                 const double* meas2, double gate_thr, double lambda_, double w_gate, int second_stage,
                                 const unsigned char* iou_ok, const double* trk_tlbr, const double* det_tlbr, double thr_embed, double thr_iou,
                                 int* match_t, int* match_d, int* n_match, int* lost_t, int* n_lost, int* new_d, int* n_new) {
    DEFT_CHECK(T >= 0 && N >= 0 && (long long)T * (long long)(N + 1) < (1ll << 31) && ld >= N, -93, "deft_associate_2d: T=%d N=%d ld=%d", T, N, ld);
    DEFT_CHECK(n_match && n_lost && n_new && (T == 0 || (mean2 && chol && gated && iou_ok && trk_tlbr && lost_t)) &&
               (N == 0 || (meas2 && det_tlbr && new_d)) && (T == 0 || N == 0 || (sim && match_t && match_d)), -93, "deft_associate_2d: null pointer");
    int nm = 0;
    std::vector<int> rows(T), cols(N), x(T), y(N), r2, c2;
    for (int t = 0; t < T; ++t) rows[t] = t;
    for (int d = 0; d < N; ++d) cols[d] = d;
    std::vector<double> cost((size_t)T * N);
    // keep the rows / columns of the current stage that stayed unmatched (ascending, like np.where(x < 0))
    auto assign = [&](double thr) {
        const int n = (int)rows.size(), m = (int)cols.size();
        r2.clear(); c2.clear();
        if (n == 0 || m == 0) return;                              // linear_assignment on an empty matrix: everything stays
        assign_limited(cost.data(), n, m, thr, x.data(), y.data());
        for (int i = 0; i < n; ++i)
            if (x[i] >= 0) { match_t[nm] = rows[i]; match_d[nm] = cols[x[i]]; ++nm; }
            else r2.push_back(rows[i]);
        for (int j = 0; j < m; ++j) if (y[j] < 0) c2.push_back(cols[j]);
        rows.swap(r2); cols.swap(c2);
    };
    // ---- stage 1 ----
    if (T && N) {
        for (int t = 0; t < T; ++t) {
            const float* st = sim + (size_t)t * ld;
            double* ct = cost.data() + (size_t)t * N;
            if (gated[t]) {
                const double l00 = chol[3 * t], l10 = chol[3 * t + 1], l11 = chol[3 * t + 2], m0 = mean2[2 * t], m1 = mean2[2 * t + 1];
                for (int d = 0; d < N; ++d) {
                    const double z0 = (meas2[2 * d] - m0) / l00;
                    const double z1 = ((meas2[2 * d + 1] - m1) - l10 * z0) / l11;
                    const double g = z0 * z0 + z1 * z1;
                    double dist = 1.0 - (double)st[d];
                    if (g > gate_thr) dist = INFINITY;
                    ct[d] = lambda_ * dist + w_gate * g;
                }
            } else {
                for (int d = 0; d < N; ++d) ct[d] = lambda_ * (1.0 - (double)st[d]) + 0.0;
            }
        }
        assign(thr_embed);
    }
    // ---- stage 2 ----
    if (second_stage && !rows.empty() && !cols.empty() && T && N) {
        const int n = (int)rows.size(), m = (int)cols.size();
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) cost[(size_t)i * m + j] = 1.0 - (double)sim[(size_t)rows[i] * ld + cols[j]];
        assign(thr_embed);
    }
    // ---- stage 3 ----
    r2.clear();
    for (int t : rows) if (iou_ok[t]) r2.push_back(t);
    rows.swap(r2);
    if (!rows.empty() && !cols.empty()) {
        const int n = (int)rows.size(), m = (int)cols.size();
        for (int i = 0; i < n; ++i) {
            const double* b = trk_tlbr + 4 * rows[i];
            const double area_b = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
            for (int j = 0; j < m; ++j) {
                const double* q = det_tlbr + 4 * cols[j];
                const double iw = std::fmin(b[2], q[2]) - std::fmax(b[0], q[0]) + 1;
                const double ih = std::fmin(b[3], q[3]) - std::fmax(b[1], q[1]) + 1;
                const double area_q = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
                const double inter = iw * ih;
                const double ua = area_b + area_q - inter;
                cost[(size_t)i * m + j] = 1.0 - ((iw > 0 && ih > 0) ? inter / ua : 0.0);
            }
        }
        assign(thr_iou);
    }
    *n_match = nm;
    *n_lost = (int)rows.size();
    for (size_t i = 0; i < rows.size(); ++i) lost_t[i] = rows[i];
    *n_new = (int)cols.size();
    for (size_t j = 0; j < cols.size(); ++j) new_d[j] = cols[j];
    return 0;
}

// The same cascade for one class of a nuScenes frame (tracker.py:850-1030 with the 3-D branches):
//   stage 0  (stage0 != 0: every class but pedestrian, :850-884) 1 - iou3d (float32, as iou_ddd_distance returns it) between the rows with `recent`
//            (seen < 3 frames ago) and every detection, linear_assignment at 0.999; the embedding stage then sees the unmatched recent rows first,
//            the older rows behind them, and the unmatched detections;
//   stage 1  lambda * (1 - sim) + 0.001 * g with the "gaussian" distance of fuse_motion_ddd (matching.py:374-415): metric 0 = the centre distance
//            sqrt(dx^2 + dy^2 + dz^2) of the LSTM filter (kalman_filter_lstm.py:92-95), metric 1 = the squared 7-component distance of the plain
//            filter (kalman_filter.py:271-273); pairs with g > max(0.2 * depth[t], gate_floor) are excluded;
//   stage 2  1 - sim on what is left;   stage 3  1 - IoU of the 2-D boxes, rows with iou_ok, threshold thr_iou (0 for nuScenes).
// trk_ddd [T][7], det_ddd [N][7] = (h, w, l, x, y, z, rot_y).
extern "C" int deft_associate_ddd(const float* sim, int ld, int T, int N, int stage0, const unsigned char* recent, const double* trk_ddd,
                                  const double* det_ddd, const double* depth, int metric, double gate_floor, double lambda_, double w_gate,
                                  const unsigned char* iou_ok, const double* trk_tlbr, const double* det_tlbr, double thr_3d, double thr_embed,
                                  double thr_iou, int* match_t, int* match_d, int* n_match, int* lost_t, int* n_lost, int* new_d, int* n_new) {
    DEFT_CHECK(T >= 0 && N >= 0 && (long long)T * (long long)(N + 1) < (1ll << 31) && ld >= N, -93, "deft_associate_ddd: T=%d N=%d ld=%d", T, N, ld);
    DEFT_CHECK(n_match && n_lost && n_new && (T == 0 || (recent && trk_ddd && depth && iou_ok && trk_tlbr && lost_t)) &&
               (N == 0 || (det_ddd && det_tlbr && new_d)) && (T == 0 || N == 0 || (sim && match_t && match_d)), -93, "deft_associate_ddd: null pointer");
    int nm = 0;
    std::vector<int> rows, older, cols(N), x(T), y(N), r2, c2;
    for (int d = 0; d < N; ++d) cols[d] = d;
    std::vector<double> cost((size_t)T * N);
    auto assign = [&](double thr) {
        const int n = (int)rows.size(), m = (int)cols.size();
        if (n == 0 || m == 0) return;
        r2.clear(); c2.clear();
        assign_limited(cost.data(), n, m, thr, x.data(), y.data());
        for (int i = 0; i < n; ++i)
            if (x[i] >= 0) { match_t[nm] = rows[i]; match_d[nm] = cols[x[i]]; ++nm; }
            else r2.push_back(rows[i]);
        for (int j = 0; j < m; ++j) if (y[j] < 0) c2.push_back(cols[j]);
        rows.swap(r2); cols.swap(c2);
    };
    // ---- stage 0 ----
    if (stage0) {
        for (int t = 0; t < T; ++t) (recent[t] ? rows : older).push_back(t);
        if (!rows.empty() && N) {
            const int n = (int)rows.size();
            std::vector<double> ct((size_t)n * 24), cd((size_t)N * 24);
            for (int i = 0; i < n; ++i) corners_of(trk_ddd + 7 * rows[i], (double (*)[3])(ct.data() + 24 * i));
            for (int d = 0; d < N; ++d) corners_of(det_ddd + 7 * d, (double (*)[3])(cd.data() + 24 * d));
            for (int i = 0; i < n; ++i)
                for (int d = 0; d < N; ++d) {
                    const float iou = (float)iou3d((const double (*)[3])(cd.data() + 24 * d), (const double (*)[3])(ct.data() + 24 * i));
                    cost[(size_t)i * N + d] = (double)(1.0f - iou);
                }
            assign(thr_3d);
        }
        rows.insert(rows.end(), older.begin(), older.end());
    } else {
        for (int t = 0; t < T; ++t) rows.push_back(t);
    }
    // ---- stage 1 ----
    if (!rows.empty() && !cols.empty()) {
        const int n = (int)rows.size(), m = (int)cols.size();
        for (int i = 0; i < n; ++i) {
            const int t = rows[i];
            const double* a = trk_ddd + 7 * t;
            const double thr = std::fmax(0.2 * depth[t], gate_floor);
            for (int j = 0; j < m; ++j) {
                const double* b = det_ddd + 7 * cols[j];
                double g;
                if (metric == 0) {
                    const double d3 = b[3] - a[3], d4 = b[4] - a[4], d5 = b[5] - a[5];
                    g = std::sqrt(d3 * d3 + d4 * d4 + d5 * d5);
                } else {
                    g = 0.0;
                    for (int q = 0; q < 7; ++q) { const double dq = b[q] - a[q]; g += dq * dq; }
                }
                double dist = 1.0 - (double)sim[(size_t)t * ld + cols[j]];
                if (g > thr) dist = INFINITY;
                cost[(size_t)i * m + j] = lambda_ * dist + w_gate * g;
            }
        }
        assign(thr_embed);
    }
    // ---- stage 2 ----
    if (!rows.empty() && !cols.empty() && T && N) {
        const int n = (int)rows.size(), m = (int)cols.size();
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) cost[(size_t)i * m + j] = 1.0 - (double)sim[(size_t)rows[i] * ld + cols[j]];
        assign(thr_embed);
    }
    // ---- stage 3 ----
    r2.clear();
    for (int t : rows) if (iou_ok[t]) r2.push_back(t);
    rows.swap(r2);
    if (!rows.empty() && !cols.empty()) {
        const int n = (int)rows.size(), m = (int)cols.size();
        for (int i = 0; i < n; ++i) {
            const double* b = trk_tlbr + 4 * rows[i];
            const double area_b = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
            for (int j = 0; j < m; ++j) {
                const double* q = det_tlbr + 4 * cols[j];
                const double iw = std::fmin(b[2], q[2]) - std::fmax(b[0], q[0]) + 1;
                const double ih = std::fmin(b[3], q[3]) - std::fmax(b[1], q[1]) + 1;
                const double area_q = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
                const double inter = iw * ih;
                const double ua = area_b + area_q - inter;
                cost[(size_t)i * m + j] = 1.0 - ((iw > 0 && ih > 0) ? inter / ua : 0.0);
            }
        }
        assign(thr_iou);
    }
    *n_match = nm;
    *n_lost = (int)rows.size();
    for (size_t i = 0; i < rows.size(); ++i) lost_t[i] = rows[i];
    *n_new = (int)cols.size();
    for (size_t j = 0; j < cols.size(); ++j) new_d[j] = cols[j];
    return 0;
}

// The DeepSORT Kalman filter of the 2-D trackers on the pool's arrays (utils/tracking_utils/kalman_filter.py), in place, host memory:
// mean [T][8] (x, y, a, h and their velocities), cov [T][8][8].
//  * deft_kf_predict = multi_predict (:165-205): F = [[I, I], [0, I]] written out as block sums (the products by 1 and 0 of np.dot are exact),
//    process noise from the height BEFORE the step -- bit for bit what the reference computes.
//  * deft_kf_update = update (:207-240) for the rows `rows[0 .. n)` with measurement meas[k] (x, y, a, h): project (:143-163), Cholesky factor of
//    the 4 x 4 innovation covariance, gain by two triangular solves (scipy cho_factor / cho_solve in the reference), mean + K (z - H mean),
//    cov - K S K^T.  Equal to the reference to round-off (another summation order than BLAS).  -94: a projected covariance that is not positive
//    definite (numpy raises LinAlgError there).
extern "C" int deft_kf_predict(double* mean, double* cov, int T) {
    DEFT_CHECK(T >= 0 && (T == 0 || (mean && cov)), -94, "deft_kf_predict: null pointer or negative size");
    const double SP = 1.0 / 20, SV = 1.0 / 160;                      // kalman_filter.py:50-51
    for (int t = 0; t < T; ++t) {
        double* m = mean + 8 * (size_t)t;
        double* P = cov + 64 * (size_t)t;
        const double h = m[3];
        const double sd[8] = {SP * h, SP * h, 1e-2 * 1.0, SP * h, SV * h, SV * h, 1e-5 * 1.0, SV * h};
        for (int j = 0; j < 4; ++j) m[j] += m[j + 4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) P[8 * i + j] += P[8 * (i + 4) + j];          // F P
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) P[8 * i + j] += P[8 * i + j + 4];            // (F P) F^T
        for (int i = 0; i < 8; ++i) P[9 * i] += sd[i] * sd[i];
    }
    return 0;
}

extern "C" int deft_kf_update(double* mean, double* cov, const int* rows, int n, const double* meas) {
    DEFT_CHECK(n >= 0 && (n == 0 || (mean && cov && rows && meas)), -94, "deft_kf_update: null pointer or negative size");
    const double SP = 1.0 / 20;
    for (int k = 0; k < n; ++k) {
        double* m = mean + 8 * (size_t)rows[k];
        double* P = cov + 64 * (size_t)rows[k];
        const double* z = meas + 4 * (size_t)k;
        const double h = m[3];
        const double sd[4] = {SP * h, SP * h, 1e-1 * 1.0, SP * h};
        double S[4][4], L[4][4] = {};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) S[i][j] = P[8 * i + j] + (i == j ? sd[i] * sd[i] : 0.0);
        for (int j = 0; j < 4; ++j) {                                // S = L L^T
            double d = S[j][j];
            for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
            DEFT_CHECK(d > 0.0 && std::isfinite(d), -94, "deft_kf_update: projected covariance of row %d is not positive definite", rows[k]);
            L[j][j] = std::sqrt(d);
            for (int i = j + 1; i < 4; ++i) {
                double v = S[i][j];
                for (int q = 0; q < j; ++q) v -= L[i][q] * L[j][q];
                L[i][j] = v / L[j][j];
            }
        }
        double X[4][8];                                              // S X = (P H^T)^T
        for (int c = 0; c < 8; ++c) {
            double y[4];
            for (int i = 0; i < 4; ++i) {
                double v = P[8 * c + i];
                for (int q = 0; q < i; ++q) v -= L[i][q] * y[q];
                y[i] = v / L[i][i];
            }
            for (int i = 3; i >= 0; --i) {
                double v = y[i];
                for (int q = i + 1; q < 4; ++q) v -= L[q][i] * X[q][c];
                X[i][c] = v / L[i][i];
            }
        }
        double innov[4], KS[8][4];
        for (int i = 0; i < 4; ++i) innov[i] = z[i] - m[i];
        for (int j = 0; j < 8; ++j) {                                // K[j][i] = X[i][j]
            double a = 0.0;
            for (int i = 0; i < 4; ++i) a += innov[i] * X[i][j];
            m[j] += a;
            for (int i = 0; i < 4; ++i) {
                double v = 0.0;
                for (int q = 0; q < 4; ++q) v += X[q][j] * S[q][i];
                KS[j][i] = v;
            }
        }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) {
                double v = 0.0;
                for (int q = 0; q < 4; ++q) v += KS[i][q] * X[q][j];
                P[8 * i + j] -= v;
            }
    }
    return 0;
}

// ---- greedy NMS of the nuScenes branch (utils/ddd_utils.py:178-245, called from detector.py:281-288 per tracking class) -----------------------
// boxes [n][4] (x1, y1, x2, y2) and scores [n] in double; candidates = the top_k highest scores (stable ascending order, as numpy's stable
// argsort -- torch.sort's order among equal scores is unspecified); repeatedly keep the best and drop what overlaps it by more than `overlap`
// (IoU = inter / ((area_j - inter) + area_i), operation for operation as deft_amd.postprocess.greedy_nms, which stays as the cross-check; a NaN
// ratio drops the box, as `nan <= overlap` is False there).  keep [n]: zero-initialised by the callee, kept indices in its first *count slots.
extern "C" int deft_greedy_nms(const double* boxes, const double* scores, int n, double overlap, int top_k, long long* keep, int* count) {
    DEFT_CHECK(n >= 0 && top_k >= 1 && keep && count && (n == 0 || (boxes && scores)), -98, "deft_greedy_nms: null pointer or bad size (n=%d top_k=%d)", n, top_k);
    for (int i = 0; i < n; ++i) keep[i] = 0;
    *count = 0;
    if (n == 0) return 0;
    std::vector<int> idx((size_t)n);
    for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return scores[a] < scores[b]; });      // (NaN scores: unordered, like numpy's sort puts them last -- never produced upstream)
    if (n > top_k) idx.erase(idx.begin(), idx.end() - top_k);
    std::vector<double> area((size_t)n);
    for (int i = 0; i < n; ++i) area[(size_t)i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    while (!idx.empty()) {
        const int i = idx.back();
        keep[(*count)++] = i;
        if (idx.size() == 1) break;
        idx.pop_back();
        size_t m = 0;
        for (size_t q = 0; q < idx.size(); ++q) {
            const int j = idx[q];
            double w = std::min(boxes[4 * j + 2], boxes[4 * i + 2]) - std::max(boxes[4 * j], boxes[4 * i]);
            double h = std::min(boxes[4 * j + 3], boxes[4 * i + 3]) - std::max(boxes[4 * j + 1], boxes[4 * i + 1]);
            w = w < 0.0 ? 0.0 : w;
            h = h < 0.0 ? 0.0 : h;
            const double inter = w * h;
            const double uni = (area[(size_t)j] - inter) + area[(size_t)i];
            if (inter / uni <= overlap) idx[m++] = j;
        }
        idx.resize(m);
    }
    return 0;
}

// ---- which nodes a track's similarity medians over, and where their rows live ------------------------------------------------------------
// STrack.get_similarity (tracker.py:221-248): the nodes younger than max_node frames; all of them while there are at most mm + 1, else the last
// mm.  The pool keeps the last L = mm + 2 nodes of every track right-aligned (newest at column L - 1), so "young" nodes form a suffix and the
// selection is the last nsel columns.  With a block table this also writes the gather table deft_track_similarity reads -- row of the node in
// the stacked affinity blocks, its block's decay, newest node first -- straight into the caller's staging buffer: one host call instead of
// ~30 numpy operations per frame (array_tracker.py _selected_nodes / _similarity, which stay as the cross-check).
extern "C" int deft_track_nodes(const long long* nf, const long long* ni, const long long* nn, int T, int L, long long fid, int mm, int max_node,
                                unsigned char* sel, const long long* blk_frame, const long long* blk_start, const long long* blk_len,
                                const float* blk_delta, int nblk, int* rows, float* scale, int* cnt, long long* bad) {
    DEFT_CHECK(T >= 0 && L >= 1 && mm >= 1 && (T == 0 || (nf && ni && nn)), -95, "deft_track_nodes: null pointer or bad size (T=%d L=%d mm=%d)", T, L, mm);
    DEFT_CHECK(rows == nullptr || (scale && cnt && (nblk == 0 || (blk_frame && blk_start && blk_len && blk_delta))), -95,
               "deft_track_nodes: the gather table needs rows, scale, cnt and the block table");
    long long f0 = 0, f1 = -1;
    std::vector<int> of;                                  // frame - f0 -> block, -1 = none
    if (rows && nblk > 0) {
        f0 = f1 = blk_frame[0];
        for (int k = 1; k < nblk; ++k) { f0 = std::min(f0, blk_frame[k]); f1 = std::max(f1, blk_frame[k]); }
        DEFT_CHECK(f1 - f0 < (1 << 20), -95, "deft_track_nodes: block frames span %lld", f1 - f0);
        of.assign((size_t)(f1 - f0 + 1), -1);
        for (int k = 0; k < nblk; ++k) of[(size_t)(blk_frame[k] - f0)] = k;
    }
    for (int t = 0; t < T; ++t) {
        const long long* f = nf + (size_t)t * L;
        const long long* id = ni + (size_t)t * L;
        const int stored = (int)std::min<long long>(nn[t], L);
        int q = 0;
        for (int c = L - stored; c < L; ++c) q += (fid - f[c] < max_node) ? 1 : 0;
        const int nsel = q <= mm + 1 ? q : mm;
        if (sel)
            for (int c = 0; c < L; ++c) sel[(size_t)t * L + c] = c >= L - nsel ? 1 : 0;
        if (!rows) continue;
        cnt[t] = nsel;
        for (int c = 0; c < L - nsel; ++c) {               // (checked in column order: the first offender is the one numpy's row-major scan reports)
            rows[(size_t)t * L + (L - 1 - c)] = 0;
            scale[(size_t)t * L + (L - 1 - c)] = 0.f;
        }
        for (int c = L - nsel; c < L; ++c) {
            const long long fr = f[c];
            const int k = (fr >= f0 && fr <= f1) ? of[(size_t)(fr - f0)] : -1;
            if (k < 0) {
                if (bad) *bad = fr;
                DEFT_CHECK(false, -96, "deft_track_nodes: no affinity block for frame %lld (track row %d)", fr, t);
            }
            if (id[c] < 0 || id[c] >= blk_len[k]) {
                if (bad) *bad = fr;
                DEFT_CHECK(false, -97, "deft_track_nodes: node id %lld outside its frame %lld (%lld rows)", id[c], fr, blk_len[k]);
            }
            rows[(size_t)t * L + (L - 1 - c)] = (int)(blk_start[k] + id[c]);
            scale[(size_t)t * L + (L - 1 - c)] = blk_delta[k];
        }
    }
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
