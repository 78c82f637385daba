/* five_point.c -- TEST INFRASTRUCTURE ONLY (see vo_oracle.h): the five-point form of the relative-pose step of
 * the bootstrap, modules/bootstrap.m:76-79 (estimateEssentialMatrix: the closed toolbox function is MSAC around a
 * five-point solver; nothing of it is in the reference, so this is a DEFINITION, N10b, parity unpinned).
 * The solver is the published algorithm of D. Nister, "An efficient solution to the five-point relative pose
 * problem", IEEE PAMI 26(6), 2004, restated from the paper:
 *   1. the 5 x 9 matrix of the epipolar constraints q1' E q0 = 0 on K^-1-normalised points; its null space X, Y, Z, W
 *      by one-sided Jacobi (fixed sweeps and pair order: the four columns of smallest norm, in column order);
 *   2. E = x X + y Y + z Z + W; det E = 0 and 2 E E' E - tr(E E') E = 0 are ten cubics in (x, y, z): a 10 x 20 matrix over
 *      the monomials [x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1];
 *   3. Gauss-Jordan elimination with partial pivoting; rows e..j (leading monomials x^2z x^2 y^2z y^2 xyz xy) give
 *      k = e - z f, l = g - z h, m = i - z j: a 3 x 3 matrix B(z) acting on (x, y, 1) with entries of degree 3, 3, 4;
 *   4. det B(z) is a polynomial of degree 10.  Its real roots by bisection between the real roots of its derivative
 *      (recursively down to the linear one) inside a Cauchy bound: every simple real root, in ascending order;
 *   5. per root: (x, y) from the best-conditioned two rows of B(z), E, and its projection on the essential manifold
 *      (singular values 1, 1, 0: vo_or_essential_project).  Up to ten matrices per sample.
 * The loop around it is essential.c's with samples of 5, every sample's solutions scanned in ascending order of z and
 * the adaptive stop (N = log(1 - p) / log(1 - w^5)) tested once per sample.
 * All arithmetic is IEEE double in a fixed operation order (-ffp-contract=off; sqrt and division only), so the HIP side
 * reproduces solutions, counts, costs and inlier sets bit for bit. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "vo_oracle.h"

void vo_or_philox_block(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t out[4]); /* p3p.c */

/* 5 distinct indices in [0, m): the first five picks of vo_or_philox_sample8's scheme */
void vo_or_philox_sample5(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
    uint32_t r[8];
    vo_or_philox_block(seed, hyp, 0, r);
    vo_or_philox_block(seed, hyp, 1, r + 4);
    uint32_t sorted[5];
    for (int j = 0; j < 5; ++j) {
        uint32_t v = r[j] % (m - (uint32_t)j);
        for (int k = 0; k < j; ++k)
            if (v >= sorted[k]) ++v;
        int pos = j;
        while (pos > 0 && sorted[pos - 1] > v) { sorted[pos] = sorted[pos - 1]; --pos; }
        sorted[pos] = v;
        idx[j] = (int32_t)v;
    }
}

/* one-sided Jacobi as in essential.c (kept separate: the two files are independent restatements of their solvers) */
static void jacobi_cols(double* A, int m, int n, double* V, int sweeps) {
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

/* monomial tables.  degree <= 1: [x y z 1]; degree <= 2: [x^2 xy xz y^2 yz z^2 x y z 1]; degree <= 3: Nister's order */
static const int M11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
static const int M21[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                               {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

/* o (degree 2, 10 coefficients) += s * a * b for two polynomials of degree 1 */
static void acc11(double* o, const double* a, const double* b, double s) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[M11[i][j]] += s * (a[i] * b[j]);
}
/* o (degree 3, 20 coefficients) += s * a (degree 2) * b (degree 1) */
static void acc21(double* o, const double* a, const double* b, double s) {
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) o[M21[i][j]] += s * (a[i] * b[j]);
}

/* value of the polynomial c[0] + c[1] z + ... + c[n] z^n */
static double horner(const double* c, int n, double z) {
    double v = c[n];
    for (int k = n - 1; k >= 0; --k) v = v * z + c[k];
    return v;
}
/* o = a * b for polynomials in z (coefficients ascending), degrees na, nb */
static void zmul(const double* a, int na, const double* b, int nb, double* o) {
    for (int k = 0; k <= na + nb; ++k) o[k] = 0.0;
    for (int i = 0; i <= na; ++i)
        for (int j = 0; j <= nb; ++j) o[i + j] += a[i] * b[j];
}

/* the simple real roots of c[0] + ... + c[n] z^n (n <= 10) inside the Cauchy bound, ascending; returns their number.
 * The roots of a polynomial lie between the roots of its derivative: from the linear derivative upwards, every
 * interval between consecutive critical points (and the bound) whose ends differ in sign is bisected to the last bit. */
static int real_roots(const double* c, int n, double* roots) {
    while (n > 0 && c[n] == 0.0) --n;
    if (n < 1) return 0;
    double bound = 0.0;
    for (int k = 0; k < n; ++k) { const double r = fabs(c[k] / c[n]); if (r > bound) bound = r; }
    bound += 1.0;
    if (!(bound < 1e12)) bound = 1e12;
    double d[11][11];                       /* d[k]: the k-th derivative, degree n - k */
    for (int i = 0; i <= n; ++i) d[0][i] = c[i];
    for (int k = 1; k < n; ++k)
        for (int i = 0; i <= n - k; ++i) d[k][i] = d[k - 1][i + 1] * (double)(i + 1);
    double prev[10], cur[10];
    int np = 0;
    for (int k = n - 1; k >= 0; --k) {      /* degree n - k: 1, 2, ..., n */
        const int deg = n - k;
        int nc = 0;
        for (int s = 0; s <= np; ++s) {
            double lo = s == 0 ? -bound : prev[s - 1], hi = s == np ? bound : prev[s];
            double flo = horner(d[k], deg, lo), fhi = horner(d[k], deg, hi);
            if (flo == 0.0) { if (s == 0) cur[nc++] = lo; continue; }      /* (an end that is a root belongs to the interval on its left) */
            if (fhi == 0.0) { cur[nc++] = hi; continue; }
            if ((flo < 0.0) == (fhi < 0.0)) continue;
            for (int it = 0; it < 200; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                const double fm = horner(d[k], deg, mid);
                if (fm == 0.0) { lo = hi = mid; break; }
                if ((fm < 0.0) == (flo < 0.0)) lo = mid; else hi = mid;
            }
            cur[nc++] = 0.5 * (lo + hi);
        }
        np = nc;
        for (int s = 0; s < nc; ++s) prev[s] = cur[s];
        if (np == 0 && deg < n) { /* no critical point: one interval, the whole bound */ }
    }
    for (int s = 0; s < np; ++s) roots[s] = prev[s];
    return np;
}

/* x0, y0, x1, y1: K^-1-normalised coordinates of 5 correspondences -> up to 10 matrices E (row-major, x1' E x0 = 0,
 * singular values 1, 1, 0), in ascending order of the root z; returns their number */
int vo_or_five_point(const double* x0, const double* y0, const double* x1, const double* y1, double* Es) {
    double A[45], V[81];
    for (int i = 0; i < 5; ++i) {
        double* r = A + 9 * i;
        r[0] = x1[i] * x0[i]; r[1] = x1[i] * y0[i]; r[2] = x1[i];
        r[3] = y1[i] * x0[i]; r[4] = y1[i] * y0[i]; r[5] = y1[i];
        r[6] = x0[i]; r[7] = y0[i]; r[8] = 1.0;
    }
    jacobi_cols(A, 5, 9, V, 16);
    /* the four columns of smallest norm, in column order */
    double nn[9]; int null_col[4], taken[9] = {0};
    for (int j = 0; j < 9; ++j) { double s = 0; for (int k = 0; k < 5; ++k) s += A[k * 9 + j] * A[k * 9 + j]; nn[j] = s; }
    for (int t = 0; t < 4; ++t) {
        int best = -1;
        for (int j = 0; j < 9; ++j) if (!taken[j] && (best < 0 || nn[j] < nn[best])) best = j;
        taken[best] = 1;
    }
    for (int j = 0, t = 0; j < 9; ++j) if (taken[j]) null_col[t++] = j;
    /* E[r][c] as a polynomial of degree 1: coefficients of [x y z 1] */
    double E1[9][4];
    for (int e = 0; e < 9; ++e)
        for (int b = 0; b < 4; ++b) E1[e][b] = V[e * 9 + null_col[b]];
    /* E E' (degree 2), its trace, the ten cubics */
    double EEt[9][10], tr[10], cons[10][20];
    memset(EEt, 0, sizeof EEt); memset(cons, 0, sizeof cons);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) acc11(EEt[3 * r + c], E1[3 * r + k], E1[3 * c + k], 1.0);
    for (int i = 0; i < 10; ++i) tr[i] = (EEt[0][i] + EEt[4][i]) + EEt[8][i];
    {   /* det E */
        double m0[10], m1[10], m2[10];
        memset(m0, 0, sizeof m0); memset(m1, 0, sizeof m1); memset(m2, 0, sizeof m2);
        acc11(m0, E1[4], E1[8], 1.0); acc11(m0, E1[5], E1[7], -1.0);
        acc11(m1, E1[3], E1[8], 1.0); acc11(m1, E1[5], E1[6], -1.0);
        acc11(m2, E1[3], E1[7], 1.0); acc11(m2, E1[4], E1[6], -1.0);
        acc21(cons[0], m0, E1[0], 1.0); acc21(cons[0], m1, E1[1], -1.0); acc21(cons[0], m2, E1[2], 1.0);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double* o = cons[1 + 3 * r + c];
            for (int k = 0; k < 3; ++k) acc21(o, EEt[3 * r + k], E1[3 * k + c], 2.0);
            acc21(o, tr, E1[3 * r + c], -1.0);
        }
    /* Gauss-Jordan on the 10 x 20 matrix, partial pivoting */
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        for (int r = col + 1; r < 10; ++r) if (fabs(cons[r][col]) > fabs(cons[piv][col])) piv = r;
        const double p = cons[piv][col];
        if (!(fabs(p) > 0.0) || !(fabs(p) < INFINITY)) return 0;
        if (piv != col)
            for (int k = 0; k < 20; ++k) { const double t = cons[piv][k]; cons[piv][k] = cons[col][k]; cons[col][k] = t; }
        for (int k = 0; k < 20; ++k) cons[col][k] = cons[col][k] / p;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = cons[r][col];
            if (f == 0.0) continue;
            for (int k = 0; k < 20; ++k) cons[r][k] = cons[r][k] - f * cons[col][k];
        }
    }
    /* B(z): rows k, l, m; entries ascending in z: x- and y-part degree 3, constant part degree 4 */
    double B[3][3][5];
    for (int q = 0; q < 3; ++q) {
        const double* e = cons[4 + 2 * q] + 10; const double* f = cons[5 + 2 * q] + 10;
        for (int part = 0; part < 2; ++part) {
            const double* pe = e + 3 * part; const double* pf = f + 3 * part;     /* [z^2 z 1] coefficients */
            B[q][part][0] = pe[2]; B[q][part][1] = pe[1] - pf[2]; B[q][part][2] = pe[0] - pf[1]; B[q][part][3] = -pf[0]; B[q][part][4] = 0.0;
        }
        B[q][2][0] = e[9]; B[q][2][1] = e[8] - f[9]; B[q][2][2] = e[7] - f[8]; B[q][2][3] = e[6] - f[7]; B[q][2][4] = -f[6];
    }
    /* det B = kx (ly m1 - l1 my) - ky (lx m1 - l1 mx) + k1 (lx my - ly mx) */
    double t1[8], t2[8], c0[8], c1[8], c2[7], d0[11], d1[11], d2[11], det[11];
    zmul(B[1][1], 3, B[2][2], 4, t1); zmul(B[1][2], 4, B[2][1], 3, t2);
    for (int k = 0; k < 8; ++k) c0[k] = t1[k] - t2[k];
    zmul(B[1][0], 3, B[2][2], 4, t1); zmul(B[1][2], 4, B[2][0], 3, t2);
    for (int k = 0; k < 8; ++k) c1[k] = t1[k] - t2[k];
    zmul(B[1][0], 3, B[2][1], 3, t1); zmul(B[1][1], 3, B[2][0], 3, t2);
    for (int k = 0; k < 7; ++k) c2[k] = t1[k] - t2[k];
    zmul(B[0][0], 3, c0, 7, d0); zmul(B[0][1], 3, c1, 7, d1); zmul(B[0][2], 4, c2, 6, d2);
    for (int k = 0; k < 11; ++k) det[k] = (d0[k] - d1[k]) + d2[k];
    for (int k = 0; k < 11; ++k) if (!(fabs(det[k]) < INFINITY)) return 0;
    double roots[10];
    const int nr = real_roots(det, 10, roots);
    int n = 0;
    for (int s = 0; s < nr; ++s) {
        const double z = roots[s];
        /* (x, y, 1) spans the null space of B(z): the cross product of two of its rows, the pair whose third component
         * -- the denominator of x and y -- is largest in magnitude (first pair wins ties) */
        double bx[3], by[3], b1[3];
        for (int q = 0; q < 3; ++q) { bx[q] = horner(B[q][0], 3, z); by[q] = horner(B[q][1], 3, z); b1[q] = horner(B[q][2], 4, z); }
        int pa = 0, pb = 1;
        double den = bx[0] * by[1] - by[0] * bx[1];
        {
            const double d02 = bx[0] * by[2] - by[0] * bx[2], d12 = bx[1] * by[2] - by[1] * bx[2];
            if (fabs(d02) > fabs(den)) { den = d02; pa = 0; pb = 2; }
            if (fabs(d12) > fabs(den)) { den = d12; pa = 1; pb = 2; }
        }
        const double x = (by[pa] * b1[pb] - b1[pa] * by[pb]) / den, y = (b1[pa] * bx[pb] - bx[pa] * b1[pb]) / den;
        double E0[9];
        int finite = 1;
        for (int e = 0; e < 9; ++e) {
            E0[e] = ((x * E1[e][0] + y * E1[e][1]) + z * E1[e][2]) + E1[e][3];
            if (!(fabs(E0[e]) < INFINITY)) finite = 0;
        }
        if (!finite) continue;
        if (vo_or_essential_project(E0, Es + 9 * n)) ++n;
    }
    return n;
}

static inline double sampson5(const double* F, double u0, double v0, double u1, double v1) {
    const double a0 = (F[0] * u0 + F[1] * v0) + F[2], a1 = (F[3] * u0 + F[4] * v0) + F[5], a2 = (F[6] * u0 + F[7] * v0) + F[8];
    const double b0 = (F[0] * u1 + F[3] * v1) + F[6], b1 = (F[1] * u1 + F[4] * v1) + F[7];
    const double num = (u1 * a0 + v1 * a1) + a2;
    const double den = (a0 * a0 + a1 * a1) + (b0 * b0 + b1 * b1);
    return (num * num) / den;
}

/* hyp_count / hyp_cost / hyp_E: 10 slots per sample (slot s of sample h at 10 h + s; unused slots: count 0, cost of m
 * outliers, E = 0) */
int vo_or_essential_ransac5(const double* p0, const double* p1, int m, const double* K, const vo_or_ransac_opts* o,
                            const int32_t* sample_idx, double* Eout, uint8_t* inliers, int* best_hyp, int* trials_used,
                            int32_t* hyp_count, uint64_t* hyp_cost, double* hyp_E) {
    if (inliers) memset(inliers, 0, (size_t)(m > 0 ? m : 0));
    if (best_hyp) *best_hyp = -1;
    if (trials_used) *trials_used = 0;
    memset(Eout, 0, 72);
    if (m < 5) return 1;
    const int H = o->num_hyp;
    const double thr = o->max_reproj;
    const uint64_t qthr = (uint64_t)(thr * 1048576.0);
    int32_t* cnt = (int32_t*)malloc(sizeof(int32_t) * 10 * (size_t)H);
    uint64_t* cost = (uint64_t*)malloc(sizeof(uint64_t) * 10 * (size_t)H);
    double* Es = (double*)calloc(90 * (size_t)H, sizeof(double));
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    int nth = o->threads > 0 ? o->threads : 1;
    (void)nth;
#pragma omp parallel for num_threads(nth) schedule(dynamic, 8)
    for (int h = 0; h < H; ++h) {
        int32_t id[5];
        int ok = 1;
        if (sample_idx) {
            memcpy(id, sample_idx + 5 * (size_t)h, sizeof id);
            for (int k = 0; k < 5; ++k) if (id[k] < 0 || id[k] >= m) ok = 0;
        } else vo_or_philox_sample5(o->seed, (uint32_t)h, (uint32_t)m, id);
        double x0[5], y0[5], x1[5], y1[5];
        int n = 0;
        double* E = Es + 90 * (size_t)h;
        if (ok) {
            for (int k = 0; k < 5; ++k) {
                const double* a = p0 + 2 * (size_t)id[k]; const double* b = p1 + 2 * (size_t)id[k];
                y0[k] = (a[1] - cy) / fy; x0[k] = ((a[0] - cx) - sk * y0[k]) / fx;
                y1[k] = (b[1] - cy) / fy; x1[k] = ((b[0] - cx) - sk * y1[k]) / fx;
            }
            n = vo_or_five_point(x0, y0, x1, y1, E);
        }
        for (int s = 0; s < 10; ++s) {
            int c_ = 0; uint64_t q = 0;
            if (s < n) {
                double F[9];
                vo_or_fundamental_from_essential(E + 9 * s, K, F);
                for (int i = 0; i < m; ++i) {
                    const double d = sampson5(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1]);
                    if (d < thr) { ++c_; q += (uint64_t)(d * 1048576.0); } else q += qthr;
                }
            } else { q = (uint64_t)m * qthr; memset(E + 9 * s, 0, 72); }
            cnt[10 * (size_t)h + s] = c_; cost[10 * (size_t)h + s] = q;
        }
    }
    int bh = -1, bcnt = 0, used = 0; uint64_t bcost = UINT64_MAX;
    for (int h = 0; h < H; ++h) {
        for (int s = 0; s < 10; ++s) {
            const size_t r = 10 * (size_t)h + s;
            if (cost[r] < bcost && cnt[r] >= o->min_inliers) { bh = (int)r; bcnt = cnt[r]; bcost = cost[r]; }
        }
        used = h + 1;
        if (o->confidence > 0) {
            double orat = 1.0 - (double)bcnt / (double)m;
            if (orat > 0.9) orat = 0.9;
            double wi = 1.0 - orat;
            double w2 = wi * wi, w4 = w2 * w2;
            double N = log(1.0 - o->confidence) / log(1.0 - w4 * wi);
            if (!(N > (double)(used + 1))) break;
        }
    }
    if (trials_used) *trials_used = used;
    if (hyp_count) memcpy(hyp_count, cnt, sizeof(int32_t) * 10 * (size_t)H);
    if (hyp_cost) memcpy(hyp_cost, cost, sizeof(uint64_t) * 10 * (size_t)H);
    if (hyp_E) memcpy(hyp_E, Es, sizeof(double) * 90 * (size_t)H);
    int status = 2;
    if (bh >= 0) {
        const double* E = Es + 9 * (size_t)bh;
        memcpy(Eout, E, 72);
        double F[9];
        vo_or_fundamental_from_essential(E, K, F);
        for (int i = 0; i < m; ++i)
            inliers[i] = (uint8_t)(sampson5(F, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1]) < thr);
        if (best_hyp) *best_hyp = bh;
        status = 0;
    }
    free(cnt); free(cost); free(Es);
    return status;
}
