/* essential.c -- TEST INFRASTRUCTURE ONLY (see vo_oracle.h): the relative-pose step of the
 * bootstrap, modules/bootstrap.m:76-79:
 *     [E, inliers] = estimateEssentialMatrix(p0, p1, cameraParams, 'MaxNumTrials', 2000,
 *                                            'Confidence', 99.5, 'MaxDistance', 0.09)
 * The toolbox function is closed (MSAC around a five-point solver); nothing of it is in the
 * reference, so this is a DEFINITION (N10), parity unpinned:
 *   sample   8 distinct correspondences per hypothesis from the Philox4x32-10 stream
 *            (counter = (hyp, 0, 0, 0) and (hyp, 1, 0, 0), key = seed), drawn one after the other
 *            from the points not drawn yet;
 *   solver   normalised eight-point algorithm on K^-1-normalised coordinates: Hartley
 *            normalisation of both samples, the 8 x 9 design matrix, its null vector by one-sided
 *            Jacobi (fixed 16 sweeps, fixed pair order), de-normalisation, projection on the
 *            essential manifold (singular values 1, 1, 0) by a second one-sided Jacobi on the 3 x 3;
 *   score    Sampson distance in pixels^2 with F = K^-T E K^-1,
 *            d = (x1' F x0)^2 / ((F x0)_1^2 + (F x0)_2^2 + (F' x1)_1^2 + (F' x1)_2^2),
 *            inlier iff d < MaxDistance; MSAC cost = sum floor(min(d, MaxDistance) * 2^20);
 *   loop     sequential semantics as the pose step (p3p.c): strictly smaller cost wins, adaptive stop
 *            with N = log(1 - p) / log(1 - w^8).
 * All decisions are taken on IEEE doubles in a fixed operation order (-ffp-contract=off), sqrt and
 * division only, so the HIP side reproduces counts, costs and inlier sets bit for bit. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "vo_oracle.h"

void vo_or_philox_block(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t out[4]); /* p3p.c */

/* 8 distinct indices in [0, m): index j is the (r_j mod (m - j))-th point not drawn before */
void vo_or_philox_sample8(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
    uint32_t r[8];
    vo_or_philox_block(seed, hyp, 0, r);
    vo_or_philox_block(seed, hyp, 1, r + 4);
    uint32_t sorted[8];
    for (int j = 0; j < 8; ++j) {
        uint32_t v = r[j] % (m - (uint32_t)j);
        for (int k = 0; k < j; ++k)      /* earlier picks in ascending order */
            if (v >= sorted[k]) ++v;
        int pos = j;
        while (pos > 0 && sorted[pos - 1] > v) { sorted[pos] = sorted[pos - 1]; --pos; }
        sorted[pos] = v;
        idx[j] = (int32_t)v;
    }
}

/* One-sided (Hestenes) Jacobi on the columns of the m x n row-major A (overwritten: A V = U S);
 * V (n x n) accumulates the rotations.  Fixed sweeps, pair order (0,1)(0,2)...(n-2,n-1). */
static void hestenes(double* A, int m, int n, double* V, int sweeps) {
    memset(V, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
    for (int s = 0; s < sweeps; ++s)
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < m; ++k) {
                    alpha += A[k * n + p] * A[k * n + p];
                    beta += A[k * n + q] * A[k * n + q];
                    gamma += A[k * n + p] * A[k * n + q];
                }
                if (gamma == 0.0) continue;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < m; ++k) {
                    double ap = A[k * n + p], aq = A[k * n + q];
                    A[k * n + p] = c * ap - sn * aq; A[k * n + q] = sn * ap + c * aq;
                }
                for (int k = 0; k < n; ++k) {
                    double vp = V[k * n + p], vq = V[k * n + q];
                    V[k * n + p] = c * vp - sn * vq; V[k * n + q] = sn * vp + c * vq;
                }
            }
}

static double colnorm2(const double* A, int m, int n, int j) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += A[k * n + j] * A[k * n + j];
    return s;
}

/* Hartley normalisation of 8 points: T = [s 0 -s cx; 0 s -s cy; 0 0 1], mean distance sqrt(2) */
static void hartley8(const double* x, const double* y, double* xn, double* yn, double* T) {
    double cx = 0, cy = 0;
    for (int i = 0; i < 8; ++i) { cx += x[i]; cy += y[i]; }
    cx /= 8.0; cy /= 8.0;
    double d = 0;
    for (int i = 0; i < 8; ++i) { double dx = x[i] - cx, dy = y[i] - cy; d += sqrt(dx * dx + dy * dy); }
    d /= 8.0;
    double s = 1.4142135623730951 / d;
    for (int i = 0; i < 8; ++i) { xn[i] = (x[i] - cx) * s; yn[i] = (y[i] - cy) * s; }
    T[0] = s; T[1] = 0; T[2] = -(s * cx); T[3] = 0; T[4] = s; T[5] = -(s * cy); T[6] = 0; T[7] = 0; T[8] = 1;
}

static void mat3(const double* A, const double* B, double* C) { /* C = A B */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}
static void mat3_tn(const double* A, const double* B, double* C) { /* C = A' B */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r] * B[c] + A[3 + r] * B[3 + c]) + A[6 + r] * B[6 + c];
}

/* nearest essential matrix (singular values 1, 1, 0) of the 3 x 3 E0; returns 0 if E0 has rank < 2 */
int vo_or_essential_project(const double* E0, double* E) {
    double A[9], V[9];
    memcpy(A, E0, sizeof A);
    hestenes(A, 3, 3, V, 16);
    double nn[3] = {colnorm2(A, 3, 3, 0), colnorm2(A, 3, 3, 1), colnorm2(A, 3, 3, 2)};
    int a = 0, b = 1, c = 2;                       /* a, b: the two largest (first wins ties) */
    if (nn[1] > nn[a]) a = 1;
    if (nn[2] > nn[a]) a = 2;
    b = a == 0 ? 1 : 0;
    for (int j = 0; j < 3; ++j) if (j != a && nn[j] > nn[b]) b = j;
    c = 3 - a - b;
    (void)c;
    if (!(nn[b] > 0.0)) { memset(E, 0, 72); return 0; }
    double sa = sqrt(nn[a]), sb = sqrt(nn[b]);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k)
            E[r * 3 + k] = (A[r * 3 + a] / sa) * V[k * 3 + a] + (A[r * 3 + b] / sb) * V[k * 3 + b];
    return 1;
}

/* x0, x1: K^-1-normalised coordinates of the 8 sampled correspondences -> E (x1' E x0 = 0) */
int vo_or_eight_point(const double* x0, const double* y0, const double* x1, const double* y1, double* E) {
    double a0[8], b0[8], a1[8], b1[8], T0[9], T1[9], A[72], V[81];
    hartley8(x0, y0, a0, b0, T0);
    hartley8(x1, y1, a1, b1, T1);
    for (int i = 0; i < 8; ++i) {
        double* r = A + 9 * i;
        r[0] = a1[i] * a0[i]; r[1] = a1[i] * b0[i]; r[2] = a1[i];
        r[3] = b1[i] * a0[i]; r[4] = b1[i] * b0[i]; r[5] = b1[i];
        r[6] = a0[i]; r[7] = b0[i]; r[8] = 1.0;
    }
    hestenes(A, 8, 9, V, 16);
    int best = 0; double bn = INFINITY;
    for (int j = 0; j < 9; ++j) { double n2 = colnorm2(A, 8, 9, j); if (n2 < bn) { bn = n2; best = j; } }
    double F[9], FT0[9], E0[9];
    for (int k = 0; k < 9; ++k) F[k] = V[k * 9 + best];
    mat3(F, T0, FT0);
    mat3_tn(T1, FT0, E0);                           /* E0 = T1' F T0 */
    int finite = 1;
    for (int k = 0; k < 9; ++k) if (!(fabs(E0[k]) < INFINITY)) finite = 0;
    if (!finite) { memset(E, 0, 72); return 0; }
    return vo_or_essential_project(E0, E);
}

/* F = K^-T E K^-1 for K = [fx s cx; 0 fy cy; 0 0 1] (row-major) */
void vo_or_fundamental_from_essential(const double* E, const double* K, double* F) {
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    /* Ki = K^-1 = [1/fx, -s/(fx fy), (s cy - cx fy)/(fx fy); 0, 1/fy, -cy/fy; 0 0 1] */
    double Ki[9] = {1.0 / fx, -sk / (fx * fy), (sk * cy - cx * fy) / (fx * fy), 0, 1.0 / fy, -cy / fy, 0, 0, 1};
    double EK[9];
    mat3(E, Ki, EK);
    mat3_tn(Ki, EK, F);
}

/* Sampson distance (pixels^2) of (u0,v0) <-> (u1,v1) under F; +inf for a degenerate denominator */
static inline double sampson(const double* F, double u0, double v0, double u1, double v1) {
    const double a0 = (F[0] * u0 + F[1] * v0) + F[2], a1 = (F[3] * u0 + F[4] * v0) + F[5], a2 = (F[6] * u0 + F[7] * v0) + F[8];
    const double b0 = (F[0] * u1 + F[3] * v1) + F[6], b1 = (F[1] * u1 + F[4] * v1) + F[7];
    const double num = (u1 * a0 + v1 * a1) + a2;
    const double den = (a0 * a0 + a1 * a1) + (b0 * b0 + b1 * b1);
    return (num * num) / den;
}

int vo_or_essential_ransac(const double* p0, const double* p1, int m, const double* K, const vo_or_ransac_opts* o,
                           const int32_t* sample_idx, double* Eout, uint8_t* inliers, int* best_hyp, int* trials_used,
                           int32_t* hyp_count, uint64_t* hyp_cost, double* hyp_E) {
    if (inliers) memset(inliers, 0, (size_t)(m > 0 ? m : 0));
    if (best_hyp) *best_hyp = -1;
    if (trials_used) *trials_used = 0;
    memset(Eout, 0, 72);
    if (m < 8) return 1;
    const int H = o->num_hyp;
    const double thr = o->max_reproj;              /* MaxDistance, pixels^2 */
    const uint64_t qthr = (uint64_t)(thr * 1048576.0);
    int32_t* cnt = (int32_t*)malloc(sizeof(int32_t) * (size_t)H);
    uint64_t* cost = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)H);
    double* Es = (double*)malloc(sizeof(double) * 9 * (size_t)H);
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    int nth = o->threads > 0 ? o->threads : 1;
    (void)nth;
#pragma omp parallel for num_threads(nth) schedule(dynamic, 8)
    for (int h = 0; h < H; ++h) {
        int32_t id[8];
        if (sample_idx) memcpy(id, sample_idx + 8 * (size_t)h, sizeof id);
        else vo_or_philox_sample8(o->seed, (uint32_t)h, (uint32_t)m, id);
        double x0[8], y0[8], x1[8], y1[8];
        for (int k = 0; k < 8; ++k) { /* K^-1 [u v 1] */
            const double* a = p0 + 2 * (size_t)id[k]; const double* b = p1 + 2 * (size_t)id[k];
            y0[k] = (a[1] - cy) / fy; x0[k] = ((a[0] - cx) - sk * y0[k]) / fx;
            y1[k] = (b[1] - cy) / fy; x1[k] = ((b[0] - cx) - sk * y1[k]) / fx;
        }
        double* E = Es + 9 * (size_t)h;
        int ok = vo_or_eight_point(x0, y0, x1, y1, E);
        int c_ = 0; uint64_t q = 0;
        if (ok) {
            double F[9];
            vo_or_fundamental_from_essential(E, K, F);
            for (int i = 0; i < m; ++i) {
                const double d = sampson(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1]);
                if (d < thr) { ++c_; q += (uint64_t)(d * 1048576.0); } else q += qthr;
            }
        } else q = (uint64_t)m * qthr;
        cnt[h] = c_; cost[h] = q;
    }
    int bh = -1, bcnt = 0, used = 0; uint64_t bcost = UINT64_MAX;
    for (int h = 0; h < H; ++h) {
        if (cost[h] < bcost && cnt[h] >= o->min_inliers) { bh = h; bcnt = cnt[h]; bcost = cost[h]; }
        used = h + 1;
        if (o->confidence > 0) {
            double orat = 1.0 - (double)bcnt / (double)m;
            if (orat > 0.9) orat = 0.9;
            double wi = 1.0 - orat;
            double w2 = wi * wi, w4 = w2 * w2;
            double N = log(1.0 - o->confidence) / log(1.0 - w4 * w4);
            if (!(N > (double)(used + 1))) break;
        }
    }
    if (trials_used) *trials_used = used;
    if (hyp_count) memcpy(hyp_count, cnt, sizeof(int32_t) * (size_t)H);
    if (hyp_cost) memcpy(hyp_cost, cost, sizeof(uint64_t) * (size_t)H);
    if (hyp_E) memcpy(hyp_E, Es, sizeof(double) * 9 * (size_t)H);
    int status = 2;
    if (bh >= 0) {
        const double* E = Es + 9 * (size_t)bh;
        memcpy(Eout, E, 72);
        double F[9];
        vo_or_fundamental_from_essential(E, K, F);
        for (int i = 0; i < m; ++i)
            inliers[i] = (uint8_t)(sampson(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1]) < thr);
        if (best_hyp) *best_hyp = bh;
        status = 0;
    }
    free(cnt); free(cost); free(Es);
    return status;
}
