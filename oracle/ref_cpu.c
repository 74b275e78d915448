/*
 * oracle/ref_cpu.c -- plain-C fp64 restatement of TrackDLO's EM registration path.
 *
 * TEST INFRASTRUCTURE ONLY (see ref_cpu.h).  PARITY PIN STATUS: PARTIAL (see ref_cpu.h).
 *
 * Follows, step for step and including the quirks:
 *   trackdlo/src/trackdlo.cpp:92-159   get_nearest_indices / calc_LLE_weights
 *   trackdlo/src/trackdlo.cpp:161-441  cpd_lle
 *   trackdlo/src/trackdlo.cpp:584-898  traverse_euclidean
 *   trackdlo/src/trackdlo.cpp:900-999  tracking_step
 *   trackdlo/src/utils.cpp:13-19       pt2pt_dis_sq / pt2pt_dis
 *   trackdlo/src/utils.cpp:172-241     isBetween / line_sphere_intersection
 * and, behind the prototype-mode switches of ref_params, utils/tracking_test.py:233-423.
 *
 * Third-party arithmetic that is NOT under /root/reference: Eigen 3.3 (CMakeLists.txt:25)
 * -- MatrixXd::inverse()/determinant() (PartialPivLU) and
 * completeOrthogonalDecomposition().solve().  Restated here as partial-pivot LU and
 * Householder QR with column pivoting; for the full-rank systems of this path all agree to
 * cond(A)*eps.  Array exp() (:298, :354) is the C library's exp here: denormals down to e^-745, zero below
 * (an all-zero Euclidean column sends the point to node 0, :310).  Eigen's SSE2 packet exp for doubles
 * clamps its argument and returns zero somewhat earlier (about e^-708.7); builds without a double packet exp
 * call std::exp.  The band between the two is not reproducible across builds of the reference itself.
 *
 * Deliberately NOT reproduced (they do not change results): per-call heap allocations of
 * pt2pt_dis by-value MatrixXd arguments, the dead P_stored copy (:299) and diff_yy (:205-212).
 * This makes the restatement FASTER than the real reference, so GPU/CPU ratios quoted against
 * it are conservative.
 */
#include "ref_cpu.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#ifdef _OPENMP
/* libref_cpu_omp.so only: number of threads of the all-cores timing column */
void ref_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------- small dense helpers */

/* Householder QR with column pivoting; A n x n col-major (overwritten), B n x nrhs. */
int ref_solve_qrcp(double *A, int n, double *B, int nrhs, double *X) {
    int *perm = (int *)malloc(sizeof(int) * (size_t)n);
    double *cn = (double *)malloc(sizeof(double) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    if (!perm || !cn || !v) { free(perm); free(cn); free(v); return -1; }
    for (int j = 0; j < n; j++) {
        perm[j] = j;
        double s = 0;
        for (int i = 0; i < n; i++) s += A[(size_t)j * n + i] * A[(size_t)j * n + i];
        cn[j] = s;
    }
    int rank = n;
    for (int k = 0; k < n; k++) {
        /* pivot column: largest remaining norm (recomputed for robustness) */
        int pj = k; double best = -1;
        for (int j = k; j < n; j++) {
            double s = 0;
            for (int i = k; i < n; i++) s += A[(size_t)j * n + i] * A[(size_t)j * n + i];
            cn[j] = s;
            if (s > best) { best = s; pj = j; }
        }
        if (pj != k) {
            for (int i = 0; i < n; i++) {
                double t = A[(size_t)k * n + i]; A[(size_t)k * n + i] = A[(size_t)pj * n + i]; A[(size_t)pj * n + i] = t;
            }
            int t = perm[k]; perm[k] = perm[pj]; perm[pj] = t;
        }
        double normx = sqrt(cn[pj]);
        if (normx == 0.0) { rank = k; break; }
        double akk = A[(size_t)k * n + k];
        double alpha = (akk > 0) ? -normx : normx;
        /* v = x - alpha e1 */
        for (int i = k; i < n; i++) v[i] = A[(size_t)k * n + i];
        v[k] -= alpha;
        double vnorm2 = 0;
        for (int i = k; i < n; i++) vnorm2 += v[i] * v[i];
        if (vnorm2 > 0) {
            for (int j = k; j < n; j++) {
                double dot = 0;
                for (int i = k; i < n; i++) dot += v[i] * A[(size_t)j * n + i];
                double f = 2.0 * dot / vnorm2;
                for (int i = k; i < n; i++) A[(size_t)j * n + i] -= f * v[i];
            }
            for (int j = 0; j < nrhs; j++) {
                double dot = 0;
                for (int i = k; i < n; i++) dot += v[i] * B[(size_t)j * n + i];
                double f = 2.0 * dot / vnorm2;
                for (int i = k; i < n; i++) B[(size_t)j * n + i] -= f * v[i];
            }
        }
        A[(size_t)k * n + k] = alpha;
        for (int i = k + 1; i < n; i++) A[(size_t)k * n + i] = 0.0;
    }
    /* back substitution on the leading rank x rank block (minimum-norm completion = 0) */
    for (int j = 0; j < nrhs; j++) {
        for (int i = 0; i < n; i++) v[i] = 0.0;
        for (int i = rank - 1; i >= 0; i--) {
            double s = B[(size_t)j * n + i];
            for (int c = i + 1; c < rank; c++) s -= A[(size_t)c * n + i] * v[c];
            v[i] = s / A[(size_t)i * n + i];
        }
        for (int i = 0; i < n; i++) X[(size_t)j * n + perm[i]] = v[i];
    }
    free(perm); free(cn); free(v);
    return rank == n ? 0 : 1;
}

/* DIAGNOSTIC ONLY (not in the reference): the same system solved in quadruple precision -- __float128 (gcc, software,
 * eps 1.9e-34) LU with partial pivoting plus two steps of iterative refinement on the double-precision A and B, rounded
 * to double at the end.  The distance between this solution and ref_solve_qrcp's is the oracle's OWN rounding error on that
 * system; the parity tests use it to size the gates of the ill-conditioned LLE systems (tests/test_solver_error.py)
 * instead of an argued constant.  Selected process-wide by ref_set_solver(1); the default (0) is the faithful QR solve
 * of :415. */
typedef __float128 xreal;
static inline xreal xabs(xreal v) { return v < 0 ? -v : v; }
static int g_solver = 0;
void ref_set_solver(int mode) { g_solver = mode; }
int ref_get_solver(void) { return g_solver; }

int ref_solve_extended(const double *A, int n, const double *B, int nrhs, double *X) {
    xreal *LU = (xreal *)malloc(sizeof(xreal) * (size_t)n * n);
    xreal *x = (xreal *)malloc(sizeof(xreal) * (size_t)n);
    xreal *r = (xreal *)malloc(sizeof(xreal) * (size_t)n);
    int *piv = (int *)malloc(sizeof(int) * (size_t)n);
    if (!LU || !x || !r || !piv) { free(LU); free(x); free(r); free(piv); return -1; }
    for (size_t i = 0; i < (size_t)n * n; i++) LU[i] = (xreal)A[i];
    int bad = 0;
    for (int k = 0; k < n; k++) {
        int p = k; xreal best = xabs(LU[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++) { xreal v = xabs(LU[(size_t)k * n + i]); if (v > best) { best = v; p = i; } }
        piv[k] = p;
        if (best == 0) { bad = 1; continue; }
        if (p != k) for (int j = 0; j < n; j++) { xreal t = LU[(size_t)j * n + k]; LU[(size_t)j * n + k] = LU[(size_t)j * n + p]; LU[(size_t)j * n + p] = t; }
        xreal inv = (xreal)1 / LU[(size_t)k * n + k];
        for (int i = k + 1; i < n; i++) LU[(size_t)k * n + i] *= inv;
        for (int j = k + 1; j < n; j++) {
            xreal u = LU[(size_t)j * n + k];
            if (u != 0) for (int i = k + 1; i < n; i++) LU[(size_t)j * n + i] -= LU[(size_t)k * n + i] * u;
        }
    }
    for (int c = 0; c < nrhs && !bad; c++) {
        for (int i = 0; i < n; i++) x[i] = 0;
        for (int pass = 0; pass < 3; pass++) {          /* pass 0 solves, passes 1..2 refine */
            for (int i = 0; i < n; i++) {
                xreal acc = (xreal)B[(size_t)c * n + i];
                for (int j = 0; j < n; j++) acc -= (xreal)A[(size_t)j * n + i] * x[j];
                r[i] = acc;
            }
            for (int k = 0; k < n; k++)                 /* the row swaps (whole rows were swapped, L included) ... */
                if (piv[k] != k) { xreal t = r[k]; r[k] = r[piv[k]]; r[piv[k]] = t; }
            for (int k = 0; k < n; k++)                 /* ... then L */
                for (int i = k + 1; i < n; i++) r[i] -= LU[(size_t)k * n + i] * r[k];
            for (int i = n - 1; i >= 0; i--) {          /* backward: U */
                xreal acc = r[i];
                for (int j = i + 1; j < n; j++) acc -= LU[(size_t)j * n + i] * r[j];
                r[i] = acc / LU[(size_t)i * n + i];
            }
            for (int i = 0; i < n; i++) x[i] += r[i];
        }
        for (int i = 0; i < n; i++) X[(size_t)c * n + i] = (double)x[i];
    }
    free(LU); free(x); free(r); free(piv);
    return bad;
}

/* partial-pivot LU inverse of a small n x n row-major matrix; returns determinant. */
static double lu_inverse_small(const double *Ain, int n, double *inv) {
    double a[12 * 12];
    int piv[12];
    memcpy(a, Ain, sizeof(double) * (size_t)(n * n));
    double det = 1.0;
    for (int i = 0; i < n; i++) piv[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); p = i; }
        if (p != k) {
            for (int j = 0; j < n; j++) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
            int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
            det = -det;
        }
        det *= a[k * n + k];
        if (a[k * n + k] == 0.0) continue;
        for (int i = k + 1; i < n; i++) {
            a[i * n + k] /= a[k * n + k];
            for (int j = k + 1; j < n; j++) a[i * n + j] -= a[i * n + k] * a[k * n + j];
        }
    }
    if (det == 0.0) return 0.0;
    for (int c = 0; c < n; c++) {
        double y[12];
        for (int i = 0; i < n; i++) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) s -= a[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = y[i];
            for (int j = i + 1; j < n; j++) s -= a[i * n + j] * inv[j * n + c];
            inv[i * n + c] = s / a[i * n + i];
        }
    }
    return det;
}

/* ---------------------------------------------------------------- LLE (trackdlo.cpp:92-159) */

static int nearest_indices(int k, int M, int idx, int extended, int *out) {
    int n = 0;
    if (!extended) {
        /* trackdlo.cpp:92-117: truncated at the ends.  For chains shorter than 2k + 1 nodes the reference clips one
         * side only and then reads Y out of bounds (undefined behaviour); the restatement clips both sides. */
        int first = idx - k, last = idx + k;
        if (idx - k < 0) first = 0;
        else if (idx + k >= M) last = M - 1;
        if (last > M - 1) last = M - 1;
        if (first < 0) first = 0;
        for (int i = first; i <= last; i++) if (i != idx) out[n++] = i;
    } else {
        /* tracking_test.py:233-247: extended on the other side */
        if (idx - k < 0) {
            for (int i = 0; i < idx; i++) out[n++] = i;
            for (int i = idx + 1; i < idx + k + 1 + abs(idx - k); i++) if (i < M) out[n++] = i;      /* (i < M: short chains) */
        } else if (idx + k >= M) {
            int last = M - 1;
            for (int i = idx - k - (idx + k - last); i < idx; i++) if (i >= 0) out[n++] = i;
            for (int i = idx + 1; i < last + 1; i++) out[n++] = i;
        } else {
            for (int i = idx - k; i < idx; i++) out[n++] = i;
            for (int i = idx + 1; i < idx + k + 1; i++) out[n++] = i;
        }
    }
    return n;
}

void ref_calc_lle_weights(int k, const double *Y, int M, int extended, double *L) {
    memset(L, 0, sizeof(double) * (size_t)M * (size_t)M);
    for (int i = 0; i < M; i++) {
        int idx[16];
        int nn = nearest_indices(k / 2, M, i, extended, idx);
        double Gi[12 * 12] = {0}, Ginv[12 * 12] = {0};
        for (int r = 0; r < nn; r++)
            for (int s = 0; s < nn; s++) {
                double acc = 0;
                for (int d = 0; d < 3; d++)
                    acc += (Y[d * M + i] - Y[d * M + idx[r]]) * (Y[d * M + i] - Y[d * M + idx[s]]);
                Gi[r * nn + s] = acc;
            }
        double det = lu_inverse_small(Gi, nn, Ginv);
        if (det == 0.0) {   /* trackdlo.cpp:139-144 */
            for (int r = 0; r < nn; r++) Gi[r * nn + r] += 0.00001;
            lu_inverse_small(Gi, nn, Ginv);
        }
        double w[12], tot = 0;
        for (int r = 0; r < nn; r++) {
            double s = 0;
            for (int c = 0; c < nn; c++) s += Ginv[r * nn + c];
            w[r] = s; tot += s;
        }
        for (int r = 0; r < nn; r++) L[(size_t)idx[r] * M + i] = w[r] / tot;   /* L(i, idx[r]) */
    }
}

/* ---------------------------------------------------------------- kernel G (trackdlo.cpp:214-233) */

void ref_kernel_G(const double *Y0, int M, double beta, int kernel, double *coord, double *G) {
    coord[0] = 0.0;
    double cur = 0;
    for (int i = 0; i < M - 1; i++) {
        double s = 0;
        for (int d = 0; d < 3; d++) { double t = Y0[d * M + i + 1] - Y0[d * M + i]; s += t * t; }
        cur += sqrt(s);
        coord[i + 1] = cur;
    }
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) {
            double g;
            if (kernel == 0) {
                double dd = fabs(coord[i] - coord[j]);
                /* 1/(2*beta * 2*beta) * exp(-sqrt(2)*d/beta) * (2*d + sqrt(2)*beta), :233 */
                g = 1.0 / (2 * beta * 2 * beta) * exp(-sqrt(2.0) * dd / beta) * (2 * dd + sqrt(2.0) * beta);
            } else if (kernel == 1) {
                double s = 0;
                for (int d = 0; d < 3; d++) { double t = Y0[d * M + i] - Y0[d * M + j]; s += t * t; }
                g = exp(-s / (2 * beta * beta));
            } else {
                double dd = fabs(coord[i] - coord[j]);
                g = exp(-(dd * dd) / (2 * beta * beta));
            }
            G[(size_t)j * M + i] = g;
        }
}

/* ---------------------------------------------------------------- cpd_lle (trackdlo.cpp:161-441) */

static inline double sqdist3(const double *Y, int M, int m, double x, double y, double z) {
    double a = Y[m] - x, b = Y[M + m] - y, c = Y[2 * M + m] - z;
    return a * a + b * b + c * c;
}

int ref_cpd_lle(const double *X_orig, int N0, double *Y, int M, double *sigma2_io,
                const ref_params *p, const double *priors, int K,
                const int *visible_nodes, int n_vis, const double *H_override,
                ref_stats *stats, ref_trace *trace) {
    if (M < 4 || N0 <= 0) return -1;
    const int D = 3;
    double sigma2 = *sigma2_io;

    /* ---- prune X (:177-195) */
    double *X = (double *)malloc(sizeof(double) * 3 * (size_t)N0);
    int *keep = (int *)malloc(sizeof(int) * (size_t)N0);
    int N = 0;
    for (int i = 0; i < N0; i++) {
        int k = 1;
        if (!p->no_prune) {
            double shortest = 100000;
            for (int j = 0; j < M; j++) {
                double dist = sqrt(sqdist3(Y, M, j, X_orig[i], X_orig[N0 + i], X_orig[2 * (size_t)N0 + i]));
                if (dist < shortest) shortest = dist;
            }
            k = shortest < 0.1;
        }
        keep[i] = k;
        N += k;
    }
    if (N == 0) { free(X); free(keep); return -2; }
    {
        int c = 0;
        for (int i = 0; i < N0; i++) if (keep[i]) {
            X[c] = X_orig[i]; X[N + c] = X_orig[N0 + i]; X[2 * (size_t)N + c] = X_orig[2 * (size_t)N0 + i];
            c++;
        }
    }
    free(keep);
    const double *Xx = X, *Xy = X + N, *Xz = X + 2 * (size_t)N;

    int converged = 1;
    size_t MM = (size_t)M * M;
    double *Y0 = (double *)malloc(sizeof(double) * 3 * M);
    memcpy(Y0, Y, sizeof(double) * 3 * M);
    double *coord = (double *)malloc(sizeof(double) * M);
    double *G = (double *)malloc(sizeof(double) * MM);
    ref_kernel_G(Y0, M, p->beta, p->kernel, coord, G);

    /* ---- LLE matrix (:236-237) */
    double *H = (double *)calloc(MM, sizeof(double));
    if (H_override) {
        memcpy(H, H_override, sizeof(double) * MM);
    } else if (p->include_lle) {   /* the reference always computes it; unused when include_lle is false */
        double *L = (double *)malloc(sizeof(double) * MM);
        ref_calc_lle_weights(6, Y0, M, p->lle_extended, L);
        /* IL = I - L ; H = IL^T IL */
        for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) L[(size_t)j * M + i] = (i == j ? 1.0 : 0.0) - L[(size_t)j * M + i];
        for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) {
            double s = 0;
            for (int k = 0; k < M; k++) s += L[(size_t)i * M + k] * L[(size_t)j * M + k];
            H[(size_t)j * M + i] = s;
        }
        free(L);
    }
    double *HG = NULL, *HY0 = NULL;
    if (p->include_lle) {
        HG = (double *)malloc(sizeof(double) * MM);
        HY0 = (double *)malloc(sizeof(double) * 3 * M);
        for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) {
            double s = 0;
            for (int k = 0; k < M; k++) s += H[(size_t)k * M + i] * G[(size_t)j * M + k];
            HG[(size_t)j * M + i] = s;
        }
        for (int i = 0; i < M; i++) for (int d = 0; d < 3; d++) {
            double s = 0;
            for (int k = 0; k < M; k++) s += H[(size_t)k * M + i] * Y0[d * M + k];
            HY0[d * M + i] = s;
        }
    }

    /* ---- J and Y_extended (:240-260) */
    double *Jd = (double *)calloc(M, sizeof(double));       /* J is a 0/1 diagonal selector */
    double *Yext = (double *)malloc(sizeof(double) * 3 * M);
    memcpy(Yext, Y0, sizeof(double) * 3 * M);
    for (int i = 0; i < K; i++) {
        int index = (int)priors[4 * i + 0];
        if (index < 0 || index >= M) { continue; }   /* reference: out-of-range row access is UB */
        Jd[index] = 1.0;
        for (int d = 0; d < 3; d++) Yext[d * M + index] = priors[4 * i + 1 + d];
    }

    /* ---- sigma2 init (:263-273) */
    if (sigma2 == 0) {
        double s = 0;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) s += sqdist3(Y0, M, i, Xx[j], Xy[j], Xz[j]);
        sigma2 = s / (double)(D * M * N);
    }

    double *Pm = (double *)malloc(sizeof(double) * (size_t)M * N);    /* (m,n) at [n*M+m] */
    double *P1 = (double *)malloc(sizeof(double) * M);
    double *PX = (double *)malloc(sizeof(double) * 3 * M);
    double *dmin = (double *)malloc(sizeof(double) * M);
    double *A = (double *)malloc(sizeof(double) * MM);
    double *B = (double *)malloc(sizeof(double) * 3 * M);
    double *W = (double *)malloc(sizeof(double) * 3 * M);
    double *T = (double *)malloc(sizeof(double) * 3 * M);
#ifdef _OPENMP
    double *col_all = (double *)malloc(sizeof(double) * 2 * M * (size_t)omp_get_max_threads());
#else
    double *col = (double *)malloc(sizeof(double) * M);
    double *geo = (double *)malloc(sizeof(double) * M);
#endif
    double *pvis = (double *)malloc(sizeof(double) * M);
    int gap_quirk = 0, it_done = 0;
    int vis_branch = (n_vis != M && n_vis != 0 && p->k_vis != 0);

    double t0 = now_s();
    for (int it = 0; it < p->max_iter; it++) {
        /* ---- distances and per-node shortest distance (:278-296) */
        for (int m = 0; m < M; m++) dmin[m] = 10000.0 * 10000.0;
#ifdef _OPENMP      /* libref_cpu_omp.so only: bench.py's all-cores column; the checker is the serial build */
#pragma omp parallel for reduction(min : dmin[:M]) schedule(static)
#endif
        for (int n = 0; n < N; n++) {
            double *c_ = Pm + (size_t)n * M;
            for (int m = 0; m < M; m++) {
                double d2 = sqdist3(Y, M, m, Xx[n], Xy[n], Xz[n]);
                c_[m] = d2;
                if (d2 < dmin[m]) dmin[m] = d2;
            }
        }
        for (int m = 0; m < M; m++) {
            double sd = sqrt(dmin[m]);          /* min of sqrt == sqrt of min (monotone, correctly rounded) */
            if (sd > 10000) sd = 10000;
            if (sd <= p->visibility_threshold) sd = 0;
            dmin[m] = sd;
        }

        /* ---- Euclidean membership, used for the per-point argmax (:298-301, :310) */
        double c = pow(2 * M_PI * sigma2, (double)D / 2) * p->mu / (1 - p->mu) * (double)M / N;
        double Np = 0;
        memset(P1, 0, sizeof(double) * M);
        memset(PX, 0, sizeof(double) * 3 * M);
        double trXtdPt1X = 0;
        if (vis_branch) {                        /* P_vis rows (:362-372) */
            double total = 0;
            for (int m = 0; m < M; m++) total += exp(-p->k_vis * dmin[m]);
            for (int m = 0; m < M; m++) pvis[m] = exp(-p->k_vis * dmin[m]) * 1.0 / total;
        }

#ifdef _OPENMP      /* per-thread scratch columns; the sums over points become per-thread partial sums (different rounding) */
#pragma omp parallel for reduction(+ : P1[:M], PX[:3 * M], trXtdPt1X, gap_quirk) schedule(static)
#endif
        for (int n = 0; n < N; n++) {
#ifdef _OPENMP
            double *col = col_all + (size_t)omp_get_thread_num() * 2 * M, *geo = col + M;
#endif
            double *c_ = Pm + (size_t)n * M;     /* holds diff_xy column */
            double colsum = 0;
            for (int m = 0; m < M; m++) { col[m] = exp(-0.5 * c_[m] / sigma2); colsum += col[m]; }
            if (p->den_guard && colsum == 0) colsum = DBL_EPSILON;
            double den = colsum + c;
            int a = 0; double best = col[0] / den;
            for (int m = 0; m < M; m++) {
                col[m] = col[m] / den;
                if (col[m] > best) { best = col[m]; a = m; }     /* maxCoeff: first maximum */
            }

            double Pt1n;
            if (p->e_mode == 1) {
                /* prototype, use_geodesic=False: P is the Euclidean membership itself */
                Pt1n = 0;
                for (int m = 0; m < M; m++) { c_[m] = col[m]; Pt1n += col[m]; }
            } else {
                int b, lo, hi;
                double da, db;
                if (p->e_mode == 0) {
                    /* ---- geodesic substitution (:313-351) */
                    int c1 = a - 1; if (c1 == -1) c1 = 2;
                    int c2 = a + 1; if (c2 == M) c2 = M - 3;
                    double d1 = sqrt(sqdist3(Y, M, c1, Xx[n], Xy[n], Xz[n]));
                    double d2_ = sqrt(sqdist3(Y, M, c2, Xx[n], Xy[n], Xz[n]));
                    b = (d1 < d2_) ? c1 : c2;
                } else {
                    /* prototype (tracking_test.py:347-355): by membership value, clamps 1 / M-2 */
                    int c1 = a - 1; if (c1 < 0) c1 = 1;
                    int c2 = a + 1; if (c2 > M - 1) c2 = M - 2;
                    b = (col[c1] > col[c2]) ? c1 : c2;
                }
                da = sqrt(sqdist3(Y, M, a, Xx[n], Xy[n], Xz[n]));
                db = sqrt(sqdist3(Y, M, b, Xx[n], Xy[n], Xz[n]));
                for (int m = 0; m < M; m++) geo[m] = 0.0;       /* fresh zero column (:305) */
                if (p->e_mode == 0) {
                    geo[a] = sqdist3(Y, M, a, Xx[n], Xy[n], Xz[n]);      /* :332 */
                    geo[b] = sqdist3(Y, M, b, Xx[n], Xy[n], Xz[n]);      /* :333 */
                    if (a < b) {
                        for (int j = 0; j < a; j++) { double t = fabs(coord[j] - coord[a]) + da; geo[j] = t * t; }
                        for (int j = b; j < M; j++) { double t = fabs(coord[j] - coord[b]) + db; geo[j] = t * t; }
                    } else {
                        for (int j = 0; j < b; j++) { double t = fabs(coord[j] - coord[b]) + db; geo[j] = t * t; }
                        for (int j = a; j < M; j++) { double t = fabs(coord[j] - coord[a]) + da; geo[j] = t * t; }
                    }
                } else {
                    /* tracking_test.py:362-372 (closed upper range on the low side) */
                    if (a < b) {
                        for (int j = 0; j <= a; j++) { double t = fabs(coord[a] - coord[j]) + da; geo[j] = t * t; }
                        for (int j = b; j < M; j++) { double t = fabs(coord[b] - coord[j]) + db; geo[j] = t * t; }
                    } else if (a > b) {
                        for (int j = 0; j <= b; j++) { double t = fabs(coord[b] - coord[j]) + db; geo[j] = t * t; }
                        for (int j = a; j < M; j++) { double t = fabs(coord[a] - coord[j]) + da; geo[j] = t * t; }
                    }
                }
                lo = a < b ? a : b; hi = a < b ? b : a;
                if (hi - lo == 2) gap_quirk++;

                /* ---- final membership (:354-383) */
                double cs = 0;
                if (vis_branch && p->e_mode == 0) {
                    for (int m = 0; m < M; m++) {
                        col[m] = exp(-0.5 * geo[m] / sigma2) * pvis[m];      /* :375 */
                        cs += col[m];
                    }
                    double c2v = pow(2 * M_PI * sigma2, (double)D / 2) * p->mu / (1 - p->mu) / N;   /* :378 */
                    den = cs + c2v;
                } else {
                    for (int m = 0; m < M; m++) { col[m] = exp(-0.5 * geo[m] / sigma2); cs += col[m]; }
                    if (p->den_guard && cs == 0) cs = DBL_EPSILON;
                    den = cs + c;
                }
                Pt1n = 0;
                for (int m = 0; m < M; m++) { c_[m] = col[m] / den; Pt1n += c_[m]; }
            }
            /* ---- reductions (:386-389) */
            for (int m = 0; m < M; m++) {
                double pv = c_[m];
                P1[m] += pv;
                PX[m] += pv * Xx[n]; PX[M + m] += pv * Xy[n]; PX[2 * M + m] += pv * Xz[n];
            }
            trXtdPt1X += Pt1n * (Xx[n] * Xx[n] + Xy[n] * Xy[n] + Xz[n] * Xz[n]);
        }
        for (int m = 0; m < M; m++) Np += P1[m];

        /* ---- M step (:392-413) */
        for (int i = 0; i < M; i++)
            for (int j = 0; j < M; j++) {
                double a_ = P1[i] * G[(size_t)j * M + i] + (i == j ? p->lambda * sigma2 : 0.0);
                if (p->include_lle) a_ += sigma2 * p->lle_weight * HG[(size_t)j * M + i];
                if (K != 0) a_ += p->alpha * Jd[i] * G[(size_t)j * M + i];
                A[(size_t)j * M + i] = a_;
            }
        for (int i = 0; i < M; i++)
            for (int d = 0; d < 3; d++) {
                double b_ = PX[d * M + i] - P1[i] * Y0[d * M + i];
                if (p->include_lle) b_ -= sigma2 * p->lle_weight * HY0[d * M + i];
                if (K != 0) b_ += p->alpha * (Yext[d * M + i] - Y0[d * M + i]);
                B[d * M + i] = b_;
            }
        if (g_solver == 1) ref_solve_extended(A, M, B, 3, W);    /* diagnostic: extended-precision solve of the same system */
        else ref_solve_qrcp(A, M, B, 3, W);          /* :415 */

        /* ---- update (:417-422) */
        for (int i = 0; i < M; i++)
            for (int d = 0; d < 3; d++) {
                double s = 0;
                for (int k = 0; k < M; k++) s += G[(size_t)k * M + i] * W[d * M + k];
                T[d * M + i] = Y0[d * M + i] + s;
            }
        double trPXtT = 0, trTtdP1T = 0;
        for (int i = 0; i < M; i++)
            for (int d = 0; d < 3; d++) {
                trPXtT += PX[d * M + i] * T[d * M + i];
                trTtdP1T += T[d * M + i] * P1[i] * T[d * M + i];
            }
        sigma2 = (trXtdPt1X - 2 * trPXtT + trTtdP1T) / (Np * D);

        double crit = 0;
        if (p->conv_rule == 0) {
            for (int i = 0; i < M; i++) {
                double s = 0;
                for (int d = 0; d < 3; d++) { double t = Y[d * M + i] - T[d * M + i]; s += t * t; }
                crit += sqrt(s);
            }
            crit /= M;                                   /* :424 */
        } else {
            for (int i = 0; i < 3 * M; i++) { double t = Y[i] - T[i]; crit += t * t; }
        }
        memcpy(Y, T, sizeof(double) * 3 * M);
        it_done = it + 1;
        if (trace) {
            if (trace->P1) memcpy(trace->P1 + (size_t)it * M, P1, sizeof(double) * M);
            if (trace->PX) memcpy(trace->PX + (size_t)it * 3 * M, PX, sizeof(double) * 3 * M);
            if (trace->Np) trace->Np[it] = Np;
            if (trace->sigma2) trace->sigma2[it] = sigma2;
            if (trace->Y) memcpy(trace->Y + (size_t)it * 3 * M, Y, sizeof(double) * 3 * M);
        }
        if (crit < p->tol) break;
        if (it == p->max_iter - 1) { converged = 0; break; }   /* :433-437 */
    }
    double t1 = now_s();

    *sigma2_io = sigma2;
    if (stats) {
        stats->iters = it_done; stats->converged = converged; stats->n_kept = N;
        stats->gap_quirk = gap_quirk; stats->loop_seconds = t1 - t0;
    }
    free(X); free(Y0); free(coord); free(G); free(H); free(HG); free(HY0); free(Jd); free(Yext);
    free(Pm); free(P1); free(PX); free(dmin); free(A); free(B); free(W); free(T); free(pvis);
#ifdef _OPENMP
    free(col_all);
#else
    free(col); free(geo);
#endif
   
    return 0;
}

/* ---------------------------------------------------------------- utils.cpp:172-241 */

static double dist3(const double *a, const double *b) {
    double s = 0;
    for (int i = 0; i < 3; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
    return sqrt(s);
}

static int is_between(const double *x, const double *a, const double *b) {
    int in_bound = 1;
    for (int i = 0; i < 3; i++) {
        if (!(a[i] - 0.0001 <= x[i] && x[i] <= b[i] + 0.0001) &&
            !(b[i] - 0.0001 <= x[i] && x[i] <= a[i] + 0.0001)) in_bound = 0;
    }
    return in_bound;
}

int ref_line_sphere_intersection(const double A[3], const double B[3], const double C[3],
                                 double radius, double out[6]) {
    int n = 0;
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < 3; i++) a += (A[i] - B[i]) * (A[i] - B[i]);
    b = 2 * ((B[0] - A[0]) * (A[0] - C[0]) + (B[1] - A[1]) * (A[1] - C[1]) + (B[2] - A[2]) * (A[2] - C[2]));
    for (int i = 0; i < 3; i++) c += (A[i] - C[i]) * (A[i] - C[i]);
    c -= radius * radius;
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    if (delta > 0) {
        double d1 = (-b + sqrt(delta)) / (2 * a);
        double d2 = (-b - sqrt(delta)) / (2 * a);
        double p1[3], p2[3];
        for (int i = 0; i < 3; i++) { p1[i] = A[i] + d1 * (B[i] - A[i]); p2[i] = A[i] + d2 * (B[i] - A[i]); }
        if (is_between(p1, A, B)) { memcpy(out + 3 * n, p1, sizeof p1); n++; }
        if (is_between(p2, A, B)) { memcpy(out + 3 * n, p2, sizeof p2); n++; }
    } else {
        double d1 = -b / (2 * a);
        double p1[3];
        for (int i = 0; i < 3; i++) p1[i] = A[i] + d1 * (B[i] - A[i]);
        if (is_between(p1, A, B)) { memcpy(out + 3 * n, p1, sizeof p1); n++; }
    }
    return n;
}

/* ---------------------------------------------------------------- traverse_euclidean (:584-898) */

static void grow(const double *g, int Mg, int i, double *o) { o[0] = g[i]; o[1] = g[Mg + i]; o[2] = g[2 * Mg + i]; }

/* one pursuit step: scans segments, returns 1 and updates (last_found, center) when an intersection is accepted.
 * dir = +1 scans i..i_end with neighbour i+1 ; dir = -1 scans downwards with neighbour i-1. */
static int pursue(const double *guide, int Mg, int dir, int i_begin, long long i_limit,
                  double look, double *center, int *last_found, double *hit) {
    for (int i = i_begin; dir > 0 ? ((long long)i + 1 <= i_limit) : ((long long)i >= i_limit); i += dir) {
        int nb = i + dir;
        if (i < 0 || i >= Mg || nb < 0 || nb >= Mg) return -1;
        double a[3], b[3], out[6];
        grow(guide, Mg, i, a); grow(guide, Mg, nb, b);
        int n = ref_line_sphere_intersection(a, b, center, look, out);
        if (n == 0) continue;
        if (n == 1 && dist3(out, b) > dist3(center, b)) continue;
        *last_found = i;
        if (n == 2) {
            if (dist3(out, b) <= dist3(out + 3, b)) memcpy(hit, out, 3 * sizeof(double));
            else memcpy(hit, out + 3, 3 * sizeof(double));
        } else memcpy(hit, out, 3 * sizeof(double));
        memcpy(center, hit, 3 * sizeof(double));
        return 1;
    }
    return 0;
}

int ref_traverse_euclidean(const double *coord, int n_coord, const double *guide, int Mg,
                           const int *vis, int n_vis, int alignment, int alignment_node_idx,
                           double *out) {
    int np = 0;
    double center[3], hit[3];
#define PUSH(idx, q) do { out[4 * np] = (double)(idx); out[4 * np + 1] = (q)[0]; out[4 * np + 2] = (q)[1]; out[4 * np + 3] = (q)[2]; np++; } while (0)
    if (n_vis <= 0 || Mg <= 0) return -1;
    if (Mg == 1) {                                   /* :590-595 */
        grow(guide, Mg, 0, center); PUSH(vis[0], center); return np;
    }
    if (alignment == 0) {                            /* :597-671 */
        grow(guide, Mg, 0, center); PUSH(vis[0], center);
        int ncons = 0;
        for (int i = 0; i < n_vis; i++) { if (i == vis[i]) ncons++; else break; }
        if (ncons == 0) return -2;                   /* size()-1 wraps in the reference -> out-of-bounds */
        int last_found = 0, seg = 0;
        while (last_found + 1 <= ncons - 1 && seg + 1 <= n_coord - 1) {
            double look = fabs(coord[seg + 1] - coord[seg]);
            int r = pursue(guide, Mg, +1, last_found, (long long)ncons - 1, look, center, &last_found, hit);
            if (r < 0) return -3;
            if (r == 0) break;
            PUSH(seg + 1, hit); seg++;
        }
    } else if (alignment == 1) {                     /* :672-748 */
        grow(guide, Mg, Mg - 1, center); PUSH(vis[n_vis - 1], center);
        int ncons = 0;
        for (int i = 1; i <= n_vis; i++) { if (vis[n_vis - i] == n_coord - i) ncons++; else break; }
        int last_found = Mg - 1, seg = n_coord - 1;
        /* unsigned comparison in the reference: (size_t)(last_found-1) >= Mg - ncons */
        while ((unsigned long long)(long long)(last_found - 1) >= (unsigned long long)((long long)Mg - ncons) && seg - 1 >= 0) {
            double look = fabs(coord[seg] - coord[seg - 1]);
            int r = pursue(guide, Mg, -1, last_found, (long long)Mg - ncons + 1, look, center, &last_found, hit);
            if (r < 0) return -3;
            if (r == 0) break;
            PUSH(seg - 1, hit); seg--;
        }
    } else {                                         /* :749-895 */
        int ai = alignment_node_idx;
        if (ai < 0 || ai >= n_vis || ai >= Mg) return -2;
        grow(guide, Mg, ai, center); PUSH(vis[ai], center);
        int ncons2 = 1;
        for (int i = ai + 1; i < n_vis; i++) { if (vis[i] - vis[i - 1] == 1) ncons2++; else break; }
        int last_found = ai, seg = vis[ai];
        while (last_found + 1 <= ai + ncons2 - 1 && seg + 1 <= n_coord - 1) {
            double look = fabs(coord[seg + 1] - coord[seg]);
            int r = pursue(guide, Mg, +1, last_found, (long long)ai + ncons2 - 1, look, center, &last_found, hit);
            if (r < 0) return -3;
            if (r == 0) break;
            PUSH(seg + 1, hit); seg++;
        }
        /* head-ward: the reference's loop increments i (:828); reading past the end of
         * visible_nodes is undefined there -- treated here as "not consecutive". */
        int ncons1 = 1;
        for (int i = ai - 1; i >= 0; i++) {
            if (i + 1 >= n_vis) break;
            if (vis[i + 1] - vis[i] == 1) ncons1++; else break;
        }
        last_found = ai; seg = vis[ai];
        grow(guide, Mg, ai, center);
        while ((unsigned long long)(long long)(last_found - 1) >= (unsigned long long)(long long)ai - (unsigned long long)ncons1 && seg - 1 >= 0) {
            double look = fabs(coord[seg] - coord[seg - 1]);
            int r = pursue(guide, Mg, -1, last_found, 1, look, center, &last_found, hit);
            if (r < 0) return -3;
            if (r == 0) break;
            PUSH(seg - 1, hit); seg--;
        }
    }
#undef PUSH
    return np;
}

/* ---------------------------------------------------------------- class state + tracking_step */

ref_tracker *ref_tracker_create(int M, double visibility_threshold, double beta, double lambda,
                                double alpha, double k_vis, double mu, int max_iter, double tol,
                                double beta_pre_proc, double lambda_pre_proc, double lle_weight) {
    ref_tracker *t = (ref_tracker *)calloc(1, sizeof(ref_tracker));
    t->M = M;
    t->Y = (double *)calloc(3 * (size_t)M, sizeof(double));
    t->guide_nodes = (double *)calloc(3 * (size_t)M, sizeof(double));
    t->Mg = M;
    t->sigma2 = 0.0;
    t->visibility_threshold = visibility_threshold;
    t->beta = beta; t->beta_pre_proc = beta_pre_proc; t->lambda = lambda; t->lambda_pre_proc = lambda_pre_proc;
    t->alpha = alpha; t->lle_weight = lle_weight; t->k_vis = k_vis; t->mu = mu; t->max_iter = max_iter; t->tol = tol;
    t->geodesic_coord = NULL; t->n_coord = 0;
    t->priors = (double *)calloc(4 * (size_t)(2 * M + 2), sizeof(double)); t->K = 0;
    return t;
}

void ref_tracker_destroy(ref_tracker *t) {
    if (!t) return;
    free(t->Y); free(t->guide_nodes); free(t->geodesic_coord); free(t->priors); free(t);
}

void ref_tracker_initialize_nodes(ref_tracker *t, const double *Y_init) {
    memcpy(t->Y, Y_init, sizeof(double) * 3 * t->M);
    memcpy(t->guide_nodes, Y_init, sizeof(double) * 3 * t->M);
    t->Mg = t->M;
}

void ref_tracker_initialize_geodesic_coord(ref_tracker *t, const double *coord, int n) {
    /* trackdlo.cpp:77-81 appends */
    t->geodesic_coord = (double *)realloc(t->geodesic_coord, sizeof(double) * (size_t)(t->n_coord + n));
    memcpy(t->geodesic_coord + t->n_coord, coord, sizeof(double) * (size_t)n);
    t->n_coord += n;
}

int ref_tracking_step(ref_tracker *t, const double *X, int N, const int *vis, int n_vis,
                      const int *vis_ext, int n_vis_ext, const double *H_pre,
                      ref_stats *stats_pre, ref_stats *stats_main) {
    int M = t->M;
    if (n_vis_ext <= 0 || n_vis_ext > M) return -1;
    t->K = 0;
    /* guide nodes = visible sub-chain (:913-921) */
    int Mg = n_vis_ext;
    t->Mg = Mg;
    if (Mg != M) {
        for (int i = 0; i < Mg; i++) for (int d = 0; d < 3; d++) t->guide_nodes[d * Mg + i] = t->Y[d * M + vis_ext[i]];
    } else {
        memcpy(t->guide_nodes, t->Y, sizeof(double) * 3 * M);
    }
    /* pre-processing registration (:925-927) */
    double sigma2_pre = t->sigma2;
    ref_params pp; memset(&pp, 0, sizeof pp);
    pp.beta = t->beta_pre_proc; pp.lambda = t->lambda_pre_proc; pp.lle_weight = t->lle_weight; pp.mu = t->mu;
    pp.max_iter = t->max_iter; pp.tol = t->tol; pp.include_lle = 1;
    pp.alpha = 0; pp.k_vis = 0; pp.visibility_threshold = 0.01;
    int rc = ref_cpd_lle(X, N, t->guide_nodes, Mg, &sigma2_pre, &pp, NULL, 0, NULL, 0, H_pre, stats_pre, NULL);
    if (rc) return rc;

    double *pv1 = (double *)malloc(sizeof(double) * 4 * (size_t)(M + 2));
    double *pv2 = (double *)malloc(sizeof(double) * 4 * (size_t)(M + 2));
    int n1, n2, ret = 0;
    if (Mg == M) {                                   /* :929-957 */
        n1 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 0, -1, pv1);
        n2 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 1, -1, pv2);
        if (n1 <= 0 || n2 <= 0) { ret = -4; goto done; }
        for (int i = 0; i < n2 / 2; i++)             /* std::reverse */
            for (int c = 0; c < 4; c++) { double tmp = pv2[4 * i + c]; pv2[4 * i + c] = pv2[4 * (n2 - 1 - i) + c]; pv2[4 * (n2 - 1 - i) + c] = tmp; }
        for (int i = 0; i < M; i++) {
            unsigned long long j2 = (unsigned long long)i - ((unsigned long long)M - (unsigned long long)n2);
            if ((double)i < pv2[0] && i < n1) {
                memcpy(t->priors + 4 * t->K, pv1 + 4 * i, 4 * sizeof(double)); t->K++;
            } else if ((double)i > pv1[4 * (n1 - 1)] && j2 < (unsigned long long)n2) {
                memcpy(t->priors + 4 * t->K, pv2 + 4 * j2, 4 * sizeof(double)); t->K++;
            } else {
                if (i >= n1 || j2 >= (unsigned long long)n2) { ret = -5; goto done; }   /* reference: out-of-bounds */
                for (int c = 0; c < 4; c++) t->priors[4 * t->K + c] = (pv1[4 * i + c] + pv2[4 * j2 + c]) / 2.0;
                t->K++;
            }
        }
    } else if (vis_ext[0] == 0 && vis_ext[n_vis_ext - 1] == M - 1) {       /* :958-967 */
        n1 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 0, -1, pv1);
        n2 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 1, -1, pv2);
        if (n1 < 0 || n2 < 0) { ret = -4; goto done; }
        memcpy(t->priors, pv1, sizeof(double) * 4 * (size_t)n1);
        memcpy(t->priors + 4 * n1, pv2, sizeof(double) * 4 * (size_t)n2);
        t->K = n1 + n2;
    } else if (vis_ext[0] == 0) {                                          /* :968-973 */
        n1 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 0, -1, pv1);
        if (n1 < 0) { ret = -4; goto done; }
        memcpy(t->priors, pv1, sizeof(double) * 4 * (size_t)n1); t->K = n1;
    } else if (vis_ext[n_vis_ext - 1] == M - 1) {                          /* :974-979 */
        n1 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 1, -1, pv1);
        if (n1 < 0) { ret = -4; goto done; }
        memcpy(t->priors, pv1, sizeof(double) * 4 * (size_t)n1); t->K = n1;
    } else {                                                               /* :980-995 */
        int ai = -1; double moved = 999999;
        for (int i = 0; i < n_vis; i++) {
            if (i >= Mg) break;      /* reference indexes guide_nodes_.row(i) with i over visible_nodes */
            double a[3], b[3];
            for (int d = 0; d < 3; d++) { a[d] = t->Y[d * M + vis[i]]; b[d] = t->guide_nodes[d * Mg + i]; }
            double dd = dist3(a, b);
            if (dd < moved) { moved = dd; ai = i; }
        }
        n1 = ref_traverse_euclidean(t->geodesic_coord, t->n_coord, t->guide_nodes, Mg, vis_ext, n_vis_ext, 2, ai, pv1);
        if (n1 < 0) { ret = -4; goto done; }
        memcpy(t->priors, pv1, sizeof(double) * 4 * (size_t)n1); t->K = n1;
    }
    {   /* main registration (:998) */
        ref_params mp; memset(&mp, 0, sizeof mp);
        mp.beta = t->beta; mp.lambda = t->lambda; mp.lle_weight = t->lle_weight; mp.mu = t->mu;
        mp.max_iter = t->max_iter; mp.tol = t->tol; mp.include_lle = 0;
        mp.alpha = t->alpha; mp.k_vis = t->k_vis; mp.visibility_threshold = t->visibility_threshold;
        ret = ref_cpd_lle(X, N, t->Y, M, &t->sigma2, &mp, t->priors, t->K, vis_ext, n_vis_ext, NULL, stats_main, NULL);
    }
done:
    free(pv1); free(pv2);
    return ret;
}

/* ---------------------------------------------------------------- trackdlo_node.cpp:257-277, :345-360 */

int ref_visibility_prepass(const double *X, int N, const double *Y, int M, double visibility_threshold, double d_vis,
                           const double *coord, double *node_dist, int *vis, int *vis_ext, int *n_ext) {
    int nv = 0, ne = 0;
    for (int m = 0; m < M; m++) {
        double shortest = 100000;                       /* :261 */
        for (int n = 0; n < N; n++) {
            double dist = sqrt(sqdist3(Y, M, m, X[n], X[(size_t)N + n], X[2 * (size_t)N + n]));
            if (dist < shortest) shortest = dist;
        }
        if (node_dist) node_dist[m] = shortest;
        if (shortest <= visibility_threshold) vis[nv++] = m;      /* :316 / :326; ascending == the sort of :346 */
    }
    if (nv > 0) {
        for (int i = 0; i < nv - 1; i++) {                           /* :351-359 */
            vis_ext[ne++] = vis[i];
            if (fabs(coord[vis[i + 1]] - coord[vis[i]]) <= d_vis)
                for (int j = 1; j < vis[i + 1] - vis[i]; j++) vis_ext[ne++] = vis[i] + j;
        }
        vis_ext[ne++] = vis[nv - 1];                                  /* :360 */
    }
    if (n_ext) *n_ext = ne;
    return nv;
}

/* ---------------------------------------------------------------- trackdlo_node.cpp:279-343: the callback's self-occlusion ("painter") test.
 * PARITY UNPINNED against OpenCV: the reference draws every edge with cv::line(..., thickness = dlo_pixel_width) into an 8-bit image and
 * looks the nodes' pixels up in it; OpenCV is absent from this image, so the thick line is restated as its geometric content -- a pixel is
 * covered when its distance from the segment between the two (integer) end pixels is at most dlo_pixel_width / 2 (OpenCV draws the
 * rectangle of that half-width around the segment plus round caps at both ends; pixels on the boundary may differ by its fixed-point
 * rounding).  Everything else follows the reference line by line: the edges' averaged camera distances (:279-284), the sort (:286-290;
 * std::sort's order of EQUAL distances is unspecified -- ascending index here), the truncating projection (:304-309), the two look-ups per
 * edge with their std::find guards (:312-334), the line drawn AFTER the look-ups (:337-341).  proj: 3 x 4 row-major.  Returns n_vis
 * (ascending, :345-346); vis needs room for M ints. */
static int px_covered(long long px, long long py, long long ax, long long ay, long long bx, long long by, long long w) {
    const long long abx = bx - ax, aby = by - ay, apx = px - ax, apy = py - ay;
    const long long ab2 = abx * abx + aby * aby, dot = apx * abx + apy * aby;
    if (ab2 == 0 || dot <= 0) return 4 * (apx * apx + apy * apy) <= w * w;
    if (dot >= ab2) { const long long bpx = px - bx, bpy = py - by; return 4 * (bpx * bpx + bpy * bpy) <= w * w; }
    const long long cr = apx * aby - apy * abx;                 /* |AP x AB|^2 / |AB|^2 = squared distance from the line */
    return 4 * cr * cr <= w * w * ab2;
}

int ref_self_occlusion(const double *Y, int M, const double *proj, int dlo_pixel_width, const double *node_dist, double visibility_threshold, int *vis) {
    int nE = M - 1, nv = 0, nd = 0;
    if (M < 2) { if (M == 1 && node_dist[0] <= visibility_threshold) vis[nv++] = 0; return nv; }
    double *avg = (double *)malloc(sizeof(double) * nE);
    int *order = (int *)malloc(sizeof(int) * nE), *drawn = (int *)malloc(sizeof(int) * nE), *col = (int *)malloc(sizeof(int) * M), *row = (int *)malloc(sizeof(int) * M);
    char *isvis = (char *)calloc(M, 1);
    for (int i = 0; i < nE; i++) {                                /* :281-284 */
        double mx = (Y[i] + Y[i + 1]) / 2, my = (Y[M + i] + Y[M + i + 1]) / 2, mz = (Y[2 * M + i] + Y[2 * M + i + 1]) / 2;
        avg[i] = sqrt(mx * mx + my * my + mz * mz); order[i] = i;
    }
    for (int i = 1; i < nE; i++) {                                /* :286-290 (insertion sort: stable) */
        int k = order[i], j = i - 1;
        while (j >= 0 && avg[order[j]] > avg[k]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = k;
    }
    for (int m = 0; m < M; m++) {                                 /* :294-297, :304-309 */
        const double h[4] = {Y[m], Y[M + m], Y[2 * M + m], 1.0};
        double u = 0, v = 0, w = 0;
        for (int k = 0; k < 4; k++) { u += proj[k] * h[k]; v += proj[4 + k] * h[k]; w += proj[8 + k] * h[k]; }
        col[m] = (int)(u / w); row[m] = (int)(v / w);
    }
    for (int e = 0; e < nE; e++) {                                /* :303-342: edges closest to the camera first */
        const int idx = order[e];
        for (int side = 0; side < 2; side++) {
            const int node = idx + side;
            int covered = 0;
            for (int d = 0; d < nd && !covered; d++)              /* projected_edges.at<uchar>(row, col) != 0 */
                covered = px_covered(col[node], row[node], col[drawn[d]], row[drawn[d]], col[drawn[d] + 1], row[drawn[d] + 1], dlo_pixel_width);
            if (!covered && node_dist[node] <= visibility_threshold) isvis[node] = 1;        /* :312-317 / :323-328 */
        }
        drawn[nd++] = idx;                                        /* :336-341 */
    }
    for (int m = 0; m < M; m++) if (isvis[m]) vis[nv++] = m;      /* :345-346 */
    free(avg); free(order); free(drawn); free(col); free(row); free(isvis);
    return nv;
}

/* ---------------------------------------------------------------- evaluator.cpp:233-283, :333-341 */

static double calc_min_distance(const double A[3], const double B[3], const double E[3]) {
    double AB[3], AE[3], cr[3];
    for (int i = 0; i < 3; i++) { AB[i] = B[i] - A[i]; AE[i] = E[i] - A[i]; }
    cr[0] = AE[1] * AB[2] - AE[2] * AB[1];                       /* utils.cpp:477-486 */
    cr[1] = -(AE[0] * AB[2] - AE[2] * AB[0]);
    cr[2] = AE[0] * AB[1] - AE[1] * AB[0];
    double nAB = sqrt(AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2]);
    double distance = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / nAB;     /* :237 */
    double dAEAB = AE[0] * AB[0] + AE[1] * AB[1] + AE[2] * AB[2];
    double dABAB = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
    double P[3], AP[3];
    for (int i = 0; i < 3; i++) { P[i] = A[i] + AB[i] * dAEAB / dABAB; AP[i] = P[i] - A[i]; }   /* :238-240 */
    double dAPAB = AP[0] * AB[0] + AP[1] * AB[1] + AP[2] * AB[2];
    if (dAPAB < 0 || dAPAB > dABAB) {                             /* :241-252 */
        double BE[3];
        for (int i = 0; i < 3; i++) BE[i] = E[i] - B[i];
        double dAE = sqrt(AE[0] * AE[0] + AE[1] * AE[1] + AE[2] * AE[2]);
        double dBE = sqrt(BE[0] * BE[0] + BE[1] * BE[1] + BE[2] * BE[2]);
        distance = (dAE > dBE) ? dBE : dAE;
    }
    return distance;
}

double ref_piecewise_error(const double *Yt, int n1, const double *Yr, int n2) {
    double total = 0.0;
    for (int idx = 0; idx < n1; idx++) {
        double dist = -1;
        double E[3] = {Yt[idx], Yt[n1 + idx], Yt[2 * n1 + idx]};
        for (int i = 0; i < n2 - 1; i++) {
            double A[3] = {Yr[i], Yr[n2 + i], Yr[2 * n2 + i]}, B[3] = {Yr[i + 1], Yr[n2 + i + 1], Yr[2 * n2 + i + 1]};
            double di = calc_min_distance(A, B, E);
            if (dist == -1 || di < dist) dist = di;
        }
        total += dist;
    }
    return total / n1;                                             /* :280 */
}

double ref_compute_error(const double *Yt, int n1, const double *Yr, int n2) {
    return (ref_piecewise_error(Yt, n1, Yr, n2) + ref_piecewise_error(Yr, n2, Yt, n1)) / 2;     /* :335-339 */
}

/* ------------------------------------------------------------------------------------------------
 * Depth image -> cloud -> voxel-grid down-sample.
 * Back-projection: trackdlo/src/trackdlo_node.cpp:195-232 (row-major pixel scan, double arithmetic,
 * result stored in the float fields of pcl::PointXYZRGB).  Down-sample: :235-241 call
 * pcl::VoxelGrid<pcl::PointXYZRGB>::filter with leaf = downsample_leaf_size on all three axes.  PCL is a
 * third-party dependency that is not under /root/reference (ROS Noetic ships PCL 1.10); this restates the
 * published algorithm of pcl/filters/impl/voxel_grid.hpp (applyFilter, downsample_all_data_ = true,
 * min_points_per_voxel_ = 0, no filter field):
 *   bounding box (getMinMax3D, float) -> min_b = floor(min * inv_leaf), div_b = max_b - min_b + 1
 *   -> cell index idx = ijk0 + ijk1 div_b0 + ijk2 div_b0 div_b1, ijk = (int)(floor(p * inv_leaf) - (float)min_b)
 *   -> sort by idx -> one output point per occupied cell in ascending idx: float sum / (float) count
 *   (CentroidPoint / AccumulatorXYZ), and `output = input` when the cell count overflows int32.
 * PARITY UNPINNED against PCL itself.  One known implementation-defined detail: PCL sorts with std::sort,
 * which is not stable, so the order in which a cell's points are added (fp32, ulp-level effect) is
 * unspecified there; this restatement adds them in input (pixel) order.
 * Returns the number of output points; X_out is column-major with leading dimension n (= return value),
 * so the caller passes a buffer of 3 * (#masked pixels) doubles.  Returns -1 on allocation failure.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { unsigned idx; int order; } vox_entry;
static int vox_cmp(const void *a, const void *b) {
    const vox_entry *p = (const vox_entry *)a, *q = (const vox_entry *)b;
    if (p->idx != q->idx) return p->idx < q->idx ? -1 : 1;
    return p->order < q->order ? -1 : (p->order > q->order);
}

int ref_depth_to_cloud(const unsigned short *depth, const unsigned char *mask, int rows, int cols,
                       double fx, double fy, double cx, double cy, double leaf_size, double *X_out, int *n_raw_out) {
    const size_t P = (size_t)rows * cols;
    float *px = (float *)malloc(sizeof(float) * 3 * (P ? P : 1));
    vox_entry *ent = (vox_entry *)malloc(sizeof(vox_entry) * (P ? P : 1));
    if (!px || !ent) { free(px); free(ent); return -1; }
    float *py = px + P, *pz = py + P;
    int n = 0;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++)
            if (mask[(size_t)i * cols + j] != 0) {                                    /* trackdlo_node.cpp:212 */
                const double pixel_x = (double)j, pixel_y = (double)i;
                const double pc_z = depth[(size_t)i * cols + j] / 1000.0;             /* :220 */
                px[n] = (float)((pixel_x - cx) * pc_z / fx);                          /* :222 */
                py[n] = (float)((pixel_y - cy) * pc_z / fy);                          /* :223 */
                pz[n] = (float)pc_z;                                                  /* :224 */
                n++;
            }
    if (n_raw_out) *n_raw_out = n;
    if (n == 0) { free(px); free(ent); return 0; }
    const float leaf = (float)leaf_size, inv = 1.0f / leaf;                           /* setLeafSize: inverse_leaf_size_ = 1 / leaf (float) */
    float mn[3] = {px[0], py[0], pz[0]}, mx[3] = {px[0], py[0], pz[0]};
    for (int k = 1; k < n; k++) {
        const float v[3] = {px[k], py[k], pz[k]};
        for (int d = 0; d < 3; d++) { if (v[d] < mn[d]) mn[d] = v[d]; if (v[d] > mx[d]) mx[d] = v[d]; }
    }
    long long dd[3];
    int min_b[3], div_b[3];
    for (int d = 0; d < 3; d++) {
        dd[d] = (long long)((mx[d] - mn[d]) * inv) + 1;
        min_b[d] = (int)floorf(mn[d] * inv);
        div_b[d] = (int)floorf(mx[d] * inv) - min_b[d] + 1;
    }
    int n_out;
    if (dd[0] * dd[1] * dd[2] > 2147483647LL) {           /* "Leaf size is too small": output = input */
        for (int k = 0; k < n; k++) { X_out[k] = px[k]; X_out[n + k] = py[k]; X_out[2 * (size_t)n + k] = pz[k]; }
        n_out = n;
    } else {
        const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
        for (int k = 0; k < n; k++) {
            const int i0 = (int)(floorf(px[k] * inv) - (float)min_b[0]);
            const int i1 = (int)(floorf(py[k] * inv) - (float)min_b[1]);
            const int i2 = (int)(floorf(pz[k] * inv) - (float)min_b[2]);
            ent[k].idx = (unsigned)(i0 + i1 * mul1 + i2 * mul2);
            ent[k].order = k;
        }
        qsort(ent, (size_t)n, sizeof(vox_entry), vox_cmp);
        n_out = 0;
        for (int k = 0; k < n; k++) if (k == 0 || ent[k].idx != ent[k - 1].idx) n_out++;
        int o = 0;
        for (int k = 0; k < n;) {
            int e = k;
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            while (e < n && ent[e].idx == ent[k].idx) { const int q = ent[e].order; sx += px[q]; sy += py[q]; sz += pz[q]; e++; }
            const float cnt = (float)(e - k);
            X_out[o] = (double)(sx / cnt); X_out[n_out + o] = (double)(sy / cnt); X_out[2 * (size_t)n_out + o] = (double)(sz / cnt);
            o++; k = e;
        }
    }
    free(px); free(ent);
    return n_out;
}

/* ------------------------------------------------------------------------------------------------
 * reg, trackdlo/src/utils.cpp:21-82: plain GMM-EM (Euclidean membership only).  pts N x 3 column-major.
 * proto = 0: the C++ function.  proto = 1: the numpy prototype `register` (utils/tracking_test.py:118-172), which
 * pins the shared maths: centroids start on the x axis (:122), sigma2 starts at 1 (:125), a zero column sum becomes
 * eps (:142), and max_iter + 1 estimates are made (:165-170).
 * ------------------------------------------------------------------------------------------------ */
void ref_reg(const double *pts, int N, double *Y, double *sigma2, int M, double mu, int max_iter, int proto) {
    const int D = 3;
    for (int i = 0; i < M; i++) {
        Y[i] = Y[M + i] = Y[2 * M + i] = 0.0;
        if (proto) Y[i] = (double)i * (0.1 / M);                                     /* np.arange(0, 0.1, 0.1 / M) */
        else Y[M + i] = 0.1 / (double)M * (double)i;                                 /* utils.cpp:26 */
    }
    double *d2 = (double *)malloc(sizeof(double) * (size_t)M * N);
    double *P = (double *)malloc(sizeof(double) * (size_t)M * N);
    double *P1 = (double *)malloc(sizeof(double) * M), *PX = (double *)malloc(sizeof(double) * 3 * M);
    if (proto) *sigma2 = 1.0;
    else {
        double s = 0.0;
        for (int n = 0; n < N; n++) for (int m = 0; m < M; m++) {                    /* :36-45 (column-major sum order) */
            const double dx = Y[m] - pts[n], dy = Y[M + m] - pts[N + n], dz = Y[2 * M + m] - pts[2 * (size_t)N + n];
            s += dx * dx + dy * dy + dz * dz;
        }
        *sigma2 = s / (double)(D * M * N);
    }
    const int iters = proto ? max_iter + 1 : max_iter;
    for (int it = 0; it < iters; it++) {
        const double s2 = *sigma2;
        const double c = pow(2 * M_PI * s2, (double)D / 2) * mu / (1 - mu) * (double)M / N;     /* :57 */
        for (int m = 0; m < M; m++) { P1[m] = 0; PX[m] = PX[M + m] = PX[2 * M + m] = 0; }
        double num = 0.0, den_s = 0.0;
        for (int n = 0; n < N; n++) {
            double den = 0.0;
            for (int m = 0; m < M; m++) {
                const double dx = Y[m] - pts[n], dy = Y[M + m] - pts[N + n], dz = Y[2 * M + m] - pts[2 * (size_t)N + n];
                const double q = dx * dx + dy * dy + dz * dz;
                d2[(size_t)n * M + m] = q;
                P[(size_t)n * M + m] = exp(-0.5 * q / s2);                           /* :55 */
                den += P[(size_t)n * M + m];
            }
            if (proto && den == 0) den = 2.220446049250313e-16;
            den += c;                                                                /* :58 */
            for (int m = 0; m < M; m++) P[(size_t)n * M + m] /= den;
        }
        for (int n = 0; n < N; n++) for (int m = 0; m < M; m++) {
            const double p = P[(size_t)n * M + m];
            P1[m] += p; PX[m] += p * pts[n]; PX[M + m] += p * pts[N + n]; PX[2 * M + m] += p * pts[2 * (size_t)N + n];
        }
        for (int m = 0; m < M; m++) for (int n = 0; n < N; n++) {                    /* :73-78 */
            num += P[(size_t)n * M + m] * d2[(size_t)n * M + m];
            den_s += P[(size_t)n * M + m] * D;
        }
        for (int m = 0; m < M; m++) { Y[m] = PX[m] / P1[m]; Y[M + m] = PX[M + m] / P1[m]; Y[2 * M + m] = PX[2 * M + m] / P1[m]; }   /* :68 */
        *sigma2 = num / den_s;                                                       /* :80 */
    }
    free(d2); free(P); free(P1); free(PX);
}
