// tdlo_host.cpp -- host-side (CPU, fp64) pieces of the path that stay on the host by design:
// they are O(M) .. O(M^2), serial and branchy.
//
//   LLE weights            trackdlo/src/trackdlo.cpp:92-159
//   line/sphere intersect  trackdlo/src/utils.cpp:172-241
//   traverse_euclidean     trackdlo/src/trackdlo.cpp:584-898
//
// Written against the behaviour of those functions, not their text: chain traversal is expressed
// as one "pursuit" routine parameterised by direction instead of four unrolled copies.
#include "tdlo_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tdlo {

// ---------------------------------------------------------------------------------------------
// LLE weights.  Local Gram matrices are at most 6 x 6; they are inverted by LU with partial
// pivoting, the algorithm Eigen's dynamic-size inverse()/determinant() use (trackdlo.cpp:136-137).
// For the 4..6-neighbour rows the Gram matrix of 3-D differences has rank <= 3, so -- exactly as in
// the reference -- the resulting weights are dominated by rounding noise; callers
// that need reproducible registrations pass H explicitly.
// ---------------------------------------------------------------------------------------------
namespace {

struct SmallLU {
    int n = 0;
    double a[36];
    int perm[6];
    double det = 1.0;

    // (the interior nodes of a chain all have six neighbours: that size gets loops with constant bounds -- the same operations in the same
    //  order, unrolled by the compiler; 45 of these factorisations sit in front of every pre-processing registration)
    void factor(const double *src, int n_) {
        if (n_ == 6) factor_n<6>(src); else factor_n<0>(src, n_);
    }
    void solve(const double *b, double *x) const {
        if (n == 6) solve_n<6>(b, x); else solve_n<0>(b, x);
    }
    template <int N> void factor_n(const double *src, int n_ = N) {
        n = n_;
        const int n = N ? N : n_;
        std::memcpy(a, src, sizeof(double) * n * n);
        det = 1.0;
        for (int i = 0; i < n; ++i) perm[i] = i;
        for (int k = 0; k < n; ++k) {
            int p = k;
            for (int i = k + 1; i < n; ++i) if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
            if (p != k) {
                for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
                std::swap(perm[k], perm[p]);
                det = -det;
            }
            const double piv = a[k * n + k];
            det *= piv;
            if (piv == 0.0) continue;
            for (int i = k + 1; i < n; ++i) {
                const double l = a[i * n + k] / piv;
                a[i * n + k] = l;
                for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
            }
        }
    }
    // solves A x = b
    template <int N> void solve_n(const double *b, double *x) const {
        const int n = N ? N : this->n;
        double y[6];
        for (int i = 0; i < n; ++i) {
            double s = b[perm[i]];
            for (int j = 0; j < i; ++j) s -= a[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * x[j];
            x[i] = s / a[i * n + i];
        }
    }
};

// (no heap: this runs M times on the host in front of every pre-processing registration -- 45 small vectors had cost half of lle_weights' 13 us)
int chain_neighbours(int half, int M, int idx, int *out, int cap) {   // trackdlo.cpp:92-117 (truncated at the ends); returns the count (may exceed cap)
    int first = idx - half, last = idx + half;
    if (idx - half < 0) first = 0;
    else if (idx + half >= M) last = M - 1;
    // Chains shorter than 2 * half + 1 nodes: the reference's if / else-if clips one side only and then indexes Y out of
    // bounds (undefined behaviour, trackdlo.cpp:96-104).  Here the range is clipped on both sides.
    if (last > M - 1) last = M - 1;
    if (first < 0) first = 0;
    int n = 0;
    for (int i = first; i <= last; ++i) if (i != idx) { if (n < cap) out[n] = i; ++n; }
    return n;
}

}  // namespace

// the weights of node i's chain neighbours (trackdlo.cpp:119-159, one row of L): returns their number n (0: none) with nb[r] / w[r]
static int lle_node_weights(int k, const double *Y, int M, int i, int *nb, double *w) {
    const int n = chain_neighbours(k / 2, M, i, nb, 6);
    if (n == 0 || n > 6) return 0;
    // local Gram matrix of the differences (trackdlo.cpp:128-134): every difference is formed once (it had been formed n times), and the matrix
    // is symmetric term by term -- (a b) and (b a) round alike, the sums run over d in the same order -- so one triangle is computed and mirrored:
    // the same 36 values, bit for bit, in a third of the operations
    double gram[36], df[6][3];
    for (int r = 0; r < n; ++r)
        for (int d = 0; d < 3; ++d) df[r][d] = Y[d * M + i] - Y[d * M + nb[r]];
    for (int r = 0; r < n; ++r)
        for (int s = r; s < n; ++s) {
            double acc = 0;
            for (int d = 0; d < 3; ++d) acc += df[r][d] * df[s][d];
            gram[r * n + s] = acc; gram[s * n + r] = acc;
        }
    SmallLU lu;
    lu.factor(gram, n);
    if (lu.det == 0.0) {                                     // trackdlo.cpp:139-144
        for (int r = 0; r < n; ++r) gram[r * n + r] += 0.00001;
        lu.factor(gram, n);
    }
    // w = Gi^-1 1 / (1^T Gi^-1 1) (:146-150); Gi^-1 1 is obtained as the solution of Gi w = 1
    double ones[6] = {1, 1, 1, 1, 1, 1}, v[6];
    lu.solve(ones, v);
    double tot = 0;
    for (int r = 0; r < n; ++r) tot += v[r];
    for (int r = 0; r < n; ++r) w[r] = v[r] / tot;
    return n;
}

void lle_weights(int k, const double *Y, int M, double *L) {
    std::fill(L, L + (size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i) {
        int nb[6];
        double w[6];
        const int n = lle_node_weights(k, Y, M, i, nb, w);
        for (int r = 0; r < n; ++r) L[(size_t)nb[r] * M + i] = w[r];
    }
}

// H = (I - L)^T (I - L) of the weights above (k = 6: +-3 chain neighbours) as its 13 diagonals only: Hb[13 i + u] = H(i, i - 6 + u), zero where
// that column does not exist.  O(M) instead of the M x M matrices of lle_weights + lle_regulariser (two 16 KB fills and a heap allocation at
// M = 45 in front of every pre-processing registration); every value is the dense routines' bit for bit -- the same weights, and the same
// products summed in the same ascending order (the terms left out there are exact zeros).
// The device forms the same 13 diagonals itself for the next frame (tdlo_lle_dev.h, `#pragma clang fp contract(off)`) and the library assumes the
// two are the same BITS (Slot::hb_next): this file must be compiled without fused multiply-adds -- the Makefile and scripts/build_variant.sh pass
// -ffp-contract=off (GCC's default is `fast`, which contracts as soon as -march / -mfma allows it).
void lle_regulariser_band(const double *Y, int M, double *Hb) {
    // A(k, c) = (I - L)(k, c) for |c - k| <= 3, stored as Ab[7 k + (c - k + 3)]
    double stack_buf[7 * 64];                                 // (no heap at production size)
    std::vector<double> heap_buf;
    double *Ab = stack_buf;
    if (M > 64) { heap_buf.assign((size_t)7 * M, 0.0); Ab = heap_buf.data(); }
    else std::fill(Ab, Ab + 7 * M, 0.0);
    for (int i = 0; i < M; ++i) {
        int nb[6];
        double w[6];
        const int n = lle_node_weights(6, Y, M, i, nb, w);
        Ab[(size_t)7 * i + 3] = 1.0;
        for (int r = 0; r < n; ++r) Ab[(size_t)7 * i + (nb[r] - i + 3)] = 0.0 - w[r];
    }
    for (int i = 0; i < M; ++i)
        for (int u = 0; u < 13; ++u) {
            const int j = i - 6 + u;
            double s = 0;
            if (j >= 0 && j < M) {
                const int k0 = std::max(0, std::max(i, j) - 3), k1 = std::min(M - 1, std::min(i, j) + 3);
                for (int k = k0; k <= k1; ++k) s += Ab[(size_t)7 * k + (i - k + 3)] * Ab[(size_t)7 * k + (j - k + 3)];
            }
            Hb[(size_t)13 * i + u] = s;
        }
}

void lle_regulariser(const double *L, int M, double *H) {        // H = (I-L)^T (I-L), trackdlo.cpp:237
    // Row k of I - L holds its diagonal 1 and the weights of k's chain neighbours (k - 3 .. k + 3, lle_weights above): entry (k, i) is zero
    // beyond |k - i| = 3, so H_ij = sum_k (I-L)_ki (I-L)_kj only has terms with k within 3 of BOTH i and j, and is zero beyond |i - j| = 6.
    // The dense triple loop (M^3 multiply-adds: 90 000 at M = 45, on the host in front of every pre-processing registration) adds exact zeros
    // for every other k; the terms that remain are summed in the same ascending order, so the values are the dense product's, bit for bit.
    std::vector<double> IL((size_t)M * M);
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i) IL[(size_t)j * M + i] = (i == j ? 1.0 : 0.0) - L[(size_t)j * M + i];
    std::fill(H, H + (size_t)M * M, 0.0);
    for (int j = 0; j < M; ++j)
        for (int i = std::max(0, j - 6); i <= std::min(M - 1, j + 6); ++i) {
            const int k0 = std::max(0, std::max(i, j) - 3), k1 = std::min(M - 1, std::min(i, j) + 3);
            double s = 0;
            for (int k = k0; k <= k1; ++k) s += IL[(size_t)i * M + k] * IL[(size_t)j * M + k];
            H[(size_t)j * M + i] = s;
        }
}

// ---------------------------------------------------------------------------------------------
// line segment / sphere intersection, utils.cpp:172-241
// ---------------------------------------------------------------------------------------------
static inline double dist(const Vec3 &a, const Vec3 &b) {
    const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

static bool within_box(const Vec3 &p, const Vec3 &a, const Vec3 &b) {   // isBetween, 0.1 mm slack per axis
    const double tol = 0.0001;
    const double pv[3] = {p.x, p.y, p.z}, av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    bool ok = true;
    for (int i = 0; i < 3; ++i) {
        const bool fwd = (av[i] - tol <= pv[i]) && (pv[i] <= bv[i] + tol);
        const bool rev = (bv[i] - tol <= pv[i]) && (pv[i] <= av[i] + tol);
        if (!fwd && !rev) ok = false;
    }
    return ok;
}

int line_sphere(const Vec3 &A, const Vec3 &B, const Vec3 &C, double radius, Vec3 out[2]) {
    const Vec3 u{B.x - A.x, B.y - A.y, B.z - A.z};
    const Vec3 w{A.x - C.x, A.y - C.y, A.z - C.z};
    const double qa = (A.x - B.x) * (A.x - B.x) + (A.y - B.y) * (A.y - B.y) + (A.z - B.z) * (A.z - B.z);
    const double qb = 2 * (u.x * w.x + u.y * w.y + u.z * w.z);
    const double qc = (w.x * w.x + w.y * w.y + w.z * w.z) - radius * radius;
    const double disc = qb * qb - 4 * qa * qc;
    int n = 0;
    if (disc < 0) return 0;
    auto at = [&](double s) { return Vec3{A.x + s * u.x, A.y + s * u.y, A.z + s * u.z}; };
    if (disc > 0) {
        const double r = std::sqrt(disc);
        const Vec3 p1 = at((-qb + r) / (2 * qa)), p2 = at((-qb - r) / (2 * qa));
        if (within_box(p1, A, B)) out[n++] = p1;
        if (within_box(p2, A, B)) out[n++] = p2;
    } else {
        const Vec3 p1 = at(-qb / (2 * qa));
        if (within_box(p1, A, B)) out[n++] = p1;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// traverse_euclidean, trackdlo.cpp:584-898: re-space the registered guide nodes at the original
// inter-node arc lengths by repeatedly intersecting a sphere (radius = next arc length) with the
// guide polyline ("pure pursuit"), starting from the head (0), the tail (1) or a middle node (2).
// ---------------------------------------------------------------------------------------------
namespace {

struct Pursuit {
    const double *guide; int Mg;
    Vec3 centre;
    int last_found;
    bool oob = false;

    Vec3 row(int i) const { return Vec3{guide[i], guide[Mg + i], guide[2 * Mg + i]}; }

    // dist(p, r) > dist(q, r) as the reference decides it (on the rounded square roots), from the squares: the square root is monotone and correctly
    // rounded, so squares that differ by more than a few ulps give roots that differ, and s1 <= s2 gives root1 <= root2; only the band between takes
    // the two square roots (sqrtsd: ~18 cycles of latency each on the path between the two registrations of tracking_step)
    static bool farther(const Vec3 &p, const Vec3 &q, const Vec3 &r) {
        const double ax = p.x - r.x, ay = p.y - r.y, az = p.z - r.z, bx = q.x - r.x, by = q.y - r.y, bz = q.z - r.z;
        const double s1 = ax * ax + ay * ay + az * az, s2 = bx * bx + by * by + bz * bz;      // (the sums dist() forms, in its order)
        if (s1 > s2 * (1.0 + 0x1p-50)) return true;
        if (s1 <= s2) return false;
        return std::sqrt(s1) > std::sqrt(s2);
    }
    // dist(p, r) <= dist(q, r), likewise (a NaN anywhere: false, as the comparison of the roots would say)
    static bool not_farther(const Vec3 &p, const Vec3 &q, const Vec3 &r) {
        const double ax = p.x - r.x, ay = p.y - r.y, az = p.z - r.z, bx = q.x - r.x, by = q.y - r.y, bz = q.z - r.z;
        const double s1 = ax * ax + ay * ay + az * az, s2 = bx * bx + by * by + bz * bz;
        if (s1 <= s2) return true;
        if (s1 > s2 * (1.0 + 0x1p-50)) return false;
        return std::sqrt(s1) <= std::sqrt(s2);
    }

    // Scans segments (i, i+dir) from last_found while `more(i)`; accepts the first usable hit.
    template <class More> bool step(int dir, double look, More more, Vec3 &hit) {
        const double lm = look - 2e-4, lm2 = lm > 0 ? lm * lm : -1.0;
        for (int i = last_found; more(i); i += dir) {
            const int j = i + dir;
            if (i < 0 || i >= Mg || j < 0 || j >= Mg) { oob = true; return false; }
            const Vec3 a = row(i), b = row(j);
            // A segment with BOTH ends more than 0.2 mm inside the sphere has no usable hit: it lies inside (convexity), the line meets the sphere beyond
            // its ends, at least look - |end - centre| > 0.2 mm beyond -- and isBetween's slack is 0.1 mm PER AXIS, which a point s beyond an end along the
            // segment exceeds on its steepest axis once s > 0.174 mm.  line_sphere would return 0 (also for a == b, where the reference's 0 / 0 fails the
            // box test): skipped without its square root, two divisions and two box tests.  This is the usual fate of the first segment of a step
            // (the one the previous hit lies on).
            {
                const double ax = a.x - centre.x, ay = a.y - centre.y, az = a.z - centre.z, bx = b.x - centre.x, by = b.y - centre.y, bz = b.z - centre.z;
                if (ax * ax + ay * ay + az * az < lm2 && bx * bx + by * by + bz * bz < lm2) continue;
            }
            Vec3 cand[2];
            const int n = line_sphere(a, b, centre, look, cand);
            if (n == 0) continue;
            if (n == 1 && farther(cand[0], centre, b)) continue;             // lone hit behind us
            last_found = i;
            hit = (n == 2 && !not_farther(cand[0], cand[1], b)) ? cand[1] : cand[0];
            centre = hit;
            return true;
        }
        return false;
    }
};

inline void push_pair(std::vector<double> &out, int idx, const Vec3 &p) {
    out.push_back((double)idx); out.push_back(p.x); out.push_back(p.y); out.push_back(p.z);
}

}  // namespace

int traverse_euclidean(const std::vector<double> &coord, const double *guide, int Mg,
                       const std::vector<int> &vis, int alignment, int anchor, std::vector<double> &out) {
    out.clear();
    const int nvis = (int)vis.size(), ncoord = (int)coord.size();
    if (nvis <= 0 || Mg <= 0) return -1;
    Pursuit pu{guide, Mg, Vec3{0, 0, 0}, 0};
    if (Mg == 1) { push_pair(out, vis[0], pu.row(0)); return 1; }               // :590-595
    Vec3 hit{0, 0, 0};

    if (alignment == 0) {                                                        // head -> tail, :597-671
        pu.centre = pu.row(0);
        push_pair(out, vis[0], pu.centre);
        int run = 0;                                   // leading run of visible nodes 0,1,2,...
        while (run < nvis && vis[run] == run) ++run;
        if (run == 0) return -1;                       // reference: size()-1 wraps, out-of-bounds reads
        pu.last_found = 0;
        int seg = 0;
        while (pu.last_found + 1 <= run - 1 && seg + 1 <= ncoord - 1) {
            const double look = std::fabs(coord[seg + 1] - coord[seg]);
            if (!pu.step(+1, look, [&](int i) { return i + 1 <= run - 1; }, hit)) break;
            push_pair(out, ++seg, hit);
        }
    } else if (alignment == 1) {                                                 // tail -> head, :672-748
        pu.centre = pu.row(Mg - 1);
        push_pair(out, vis[nvis - 1], pu.centre);
        int run = 0;                                   // trailing run ..., ncoord-2, ncoord-1
        while (run < nvis && vis[nvis - 1 - run] == ncoord - 1 - run) ++run;
        pu.last_found = Mg - 1;
        int seg = ncoord - 1;
        const long long floor_row = (long long)Mg - run;     // lowest guide row that belongs to the run
        // the reference compares (size_t)(last_found-1) >= Mg-run: a wrapped -1 passes, then the
        // inner scan is empty and the loop ends -- same outcome as stopping here.
        while ((pu.last_found - 1 >= floor_row || pu.last_found - 1 < 0) && seg - 1 >= 0) {
            const double look = std::fabs(coord[seg] - coord[seg - 1]);
            if (!pu.step(-1, look, [&](int i) { return (long long)i >= floor_row + 1; }, hit)) break;
            push_pair(out, --seg, hit);
        }
    } else {                                                                     // from a middle anchor, :749-895
        if (anchor < 0 || anchor >= nvis || anchor >= Mg) return -1;
        pu.centre = pu.row(anchor);
        push_pair(out, vis[anchor], pu.centre);
        int fwd = 1;                                   // consecutive run starting at the anchor
        while (anchor + fwd < nvis && vis[anchor + fwd] - vis[anchor + fwd - 1] == 1) ++fwd;
        pu.last_found = anchor;
        int seg = vis[anchor];
        while (pu.last_found + 1 <= anchor + fwd - 1 && seg + 1 <= ncoord - 1) {
            const double look = std::fabs(coord[seg + 1] - coord[seg]);
            if (!pu.step(+1, look, [&](int i) { return i + 1 <= anchor + fwd - 1; }, hit)) break;
            push_pair(out, ++seg, hit);
        }
        // Head-ward part.  The reference counts this run with a loop that walks TOWARDS THE TAIL
        // (its index is incremented, :828) and stops at the first non-consecutive pair; reading past
        // the end of visible_nodes is undefined there and is taken as "stop" here.  The resulting
        // count enters an unsigned comparison (:842): when it exceeds the anchor index the
        // subtraction wraps and the head-ward pursuit is skipped altogether.
        int back = 1;
        if (anchor - 1 >= 0)
            for (int i = anchor - 1; i + 1 < nvis && vis[i + 1] - vis[i] == 1; ++i) ++back;
        pu.last_found = anchor;
        seg = vis[anchor];
        pu.centre = pu.row(anchor);
        const bool wraps = back > anchor;              // (size_t)(anchor - back) is astronomically large
        auto gate = [&]() {
            if (wraps) return pu.last_found - 1 < 0;   // only a wrapped left-hand side can pass
            return pu.last_found - 1 < 0 || pu.last_found - 1 >= anchor - back;
        };
        while (gate() && seg - 1 >= 0) {
            const double look = std::fabs(coord[seg] - coord[seg - 1]);
            if (!pu.step(-1, look, [&](int i) { return i - 1 >= 0; }, hit)) break;
            push_pair(out, --seg, hit);
        }
    }
    if (pu.oob) return -1;
    return (int)(out.size() / 4);
}

// ---------------------------------------------------------------------------------------------
// Frame-level accuracy metric of the reference's evaluator:
// mean distance from each node of one chain to the other chain's polyline
// (evaluator::calc_min_distance / get_piecewise_error, trackdlo/src/evaluator.cpp:233-283) and its
// symmetrised form (compute_error, :333-341).  O(M^2) on the host.
// ---------------------------------------------------------------------------------------------
static double point_segment_distance(const Vec3 &A, const Vec3 &B, const Vec3 &E) {
    const Vec3 ab{B.x - A.x, B.y - A.y, B.z - A.z}, ae{E.x - A.x, E.y - A.y, E.z - A.z};
    const Vec3 cr{ae.y * ab.z - ae.z * ab.y, -(ae.x * ab.z - ae.z * ab.x), ae.x * ab.y - ae.y * ab.x};
    const double ab2 = ab.x * ab.x + ab.y * ab.y + ab.z * ab.z;
    double d = std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z) / std::sqrt(ab2);      // distance to the infinite line
    const double tpar = (ae.x * ab.x + ae.y * ab.y + ae.z * ab.z) / ab2;                   // foot point parameter
    const Vec3 ap{ab.x * tpar, ab.y * tpar, ab.z * tpar};
    const double proj = ap.x * ab.x + ap.y * ab.y + ap.z * ab.z;
    if (proj < 0 || proj > ab2) {                       // foot point outside the segment: nearer end point
        const double da = std::sqrt(ae.x * ae.x + ae.y * ae.y + ae.z * ae.z);
        const Vec3 be{E.x - B.x, E.y - B.y, E.z - B.z};
        const double db = std::sqrt(be.x * be.x + be.y * be.y + be.z * be.z);
        d = (da > db) ? db : da;
    }
    return d;
}

double piecewise_error(const double *Ytrack, int n1, const double *Ytrue, int n2) {
    double total = 0.0;
    for (int i = 0; i < n1; ++i) {
        const Vec3 E{Ytrack[i], Ytrack[n1 + i], Ytrack[2 * n1 + i]};
        double best = -1;
        for (int j = 0; j + 1 < n2; ++j) {
            const Vec3 A{Ytrue[j], Ytrue[n2 + j], Ytrue[2 * n2 + j]}, B{Ytrue[j + 1], Ytrue[n2 + j + 1], Ytrue[2 * n2 + j + 1]};
            const double d = point_segment_distance(A, B, E);
            if (best == -1 || d < best) best = d;
        }
        total += best;
    }
    return total / n1;
}

// ---- trackdlo_node.cpp:279-343: which nodes does the rope itself hide from the camera? ------------------------------------------------------------
// The reference paints the edges into an image nearest first (cv::line, thickness dlo_pixel_width) and looks every node's pixel up just before an edge
// it ends is painted.  A node is looked up twice at most -- once per incident edge -- and the second look-up always finds its first incident edge
// painted over its own pixel: what decides is the FIRST look-up, against the edges painted before the nearer of the node's incident edges.  So, per
// node: rank of its nearer incident edge in the painting order, then "does any edge of smaller rank cover my pixel" -- O(M^2) integer tests on the
// host, no image.  Coverage is the geometric content of the thick line (within width / 2 of the segment between the end pixels, in exact integer
// arithmetic); OpenCV's own rasteriser is not available here to pin its boundary pixels against.
namespace {
struct Px { long long c, r; };
inline bool within_half_width(const Px &p, const Px &a, const Px &b, long long w) {
    const long long ex = b.c - a.c, ey = b.r - a.r, fx = p.c - a.c, fy = p.r - a.r;
    const long long len2 = ex * ex + ey * ey, along = fx * ex + fy * ey;
    if (along <= 0 || len2 == 0) return 4 * (fx * fx + fy * fy) <= w * w;                 // in front of the first end pixel (or a zero-length edge): its cap
    if (along >= len2) { const long long gx = p.c - b.c, gy = p.r - b.r; return 4 * (gx * gx + gy * gy) <= w * w; }
    const long long area = fx * ey - fy * ex;                                               // twice the triangle's area: distance = |area| / len
    return 4 * area * area <= w * w * len2;
}
}  // namespace

void self_occlusion_visible(const double *Y, int M, const double proj[12], int dlo_pixel_width, const double *node_dist, double visibility_threshold, std::vector<int> &vis) {
    vis.clear();
    if (M < 1) return;
    if (M == 1) { if (node_dist[0] <= visibility_threshold) vis.push_back(0); return; }
    const int nE = M - 1;
    std::vector<Px> px(M);
    for (int m = 0; m < M; ++m) {                                   // :294-297, :304-309: homogeneous projection, truncated like static_cast<int>
        const double x = Y[m], y = Y[M + m], z = Y[2 * (size_t)M + m];
        const double u = ((proj[0] * x + proj[1] * y) + proj[2] * z) + proj[3] * 1.0, v = ((proj[4] * x + proj[5] * y) + proj[6] * z) + proj[7] * 1.0,
                     w = ((proj[8] * x + proj[9] * y) + proj[10] * z) + proj[11] * 1.0;
        px[m].c = (long long)(int)(u / w); px[m].r = (long long)(int)(v / w);
    }
    std::vector<std::pair<double, int>> key(nE);
    for (int i = 0; i < nE; ++i) {                                  // :281-284
        const double mx = (Y[i] + Y[i + 1]) / 2, my = (Y[M + i] + Y[M + i + 1]) / 2, mz = (Y[2 * (size_t)M + i] + Y[2 * (size_t)M + i + 1]) / 2;
        key[i] = {std::sqrt(mx * mx + my * my + mz * mz), i};
    }
    std::sort(key.begin(), key.end());                              // :286-290 (equal distances: ascending index)
    std::vector<int> rank(nE);
    for (int r = 0; r < nE; ++r) rank[key[r].second] = r;
    for (int m = 0; m < M; ++m) {
        if (!(node_dist[m] <= visibility_threshold)) continue;      // :313 / :324
        const int first = (m == 0) ? rank[0] : (m == M - 1 ? rank[nE - 1] : std::min(rank[m - 1], rank[m]));
        bool hidden = false;
        for (int r = 0; r < first && !hidden; ++r) { const int e = key[r].second; hidden = within_half_width(px[m], px[e], px[e + 1], dlo_pixel_width); }
        if (!hidden) vis.push_back(m);
    }
}

}  // namespace tdlo
