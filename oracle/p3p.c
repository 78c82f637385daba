/* oracle/p3p.c -- CPU restatement of P3P-RANSAC localisation (TEST INFRASTRUCTURE).
 *
 * Reference (open MATLAB, restated line by line): old/solveQuartic.m:26-66 (Ferrari, complex
 * arithmetic), old/p3p.m:41-217 (Kneip 2011), old/ransacLocalization.m:44-130 (RANSAC loop).
 * The live pose step (modules/VisualOdometry.m:285-289) calls the closed-source
 * estimateWorldCameraPose (P3P inside MSAC, 5000 trials, 99 %, 2 px); its MSAC cost is DEFINED
 * here (N9): cost = sum_i floor(min(e_i^2, thr^2) * 2^20) as uint64 (order-independent, exact).
 *
 * N9 details
 *  - sample stream: hypothesis h draws 3 distinct indices from Philox4x32-10(counter=(h,0,0,0),
 *    key=(seed_lo,seed_hi)): i0=r0%m; i1=r1%(m-1) (+1 if >=i0); i2=r2%(m-2) (+1 per smaller-or-
 *    equal already-drawn index, ascending).  Replaces datasample (ransacLocalization.m:46-47).
 *  - bearings K\[u;v;1] by back-substitution on upper-triangular K, normalised (:52-56).
 *  - decode R_C_W = R^T, t_C_W = -R^T C (:63-68); project; e^2 < thr^2 strict (:82-83).
 *  - best of the 4 solutions: strict >, ties keep earlier (:86-99).
 *  - best-so-far update strict > and >= min_inliers (:107-111) [mode 0]; strict < on cost [mode 1].
 *  - adaptive stop (:113-124) with the option values instead of the hard-coded 0.95 / 15000:
 *    or = min(0.9, 1 - best/m); N = log(1-conf)/log(1-(1-or)^3); after `used` hypotheses the
 *    loop goes on only while N > used + 1 (:44, `while num_iterations > i`) and used < num_hyp.
 *  - the result is the best hypothesis' pose (no DLT refit: estimatePoseDLT is not in the repo,
 *    ransacLocalization.m:136-140).
 *  - complex principal cube root (solveQuartic.m:51,54): |z|^(1/3) (cos(arg/3) + i sin(arg/3)).
 */
#include "vo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cpx;
static inline cpx cmk(double a, double b) { cpx z = {a, b}; return z; }
static inline cpx cadd(cpx a, cpx b) { return cmk(a.re + b.re, a.im + b.im); }
static inline cpx csub(cpx a, cpx b) { return cmk(a.re - b.re, a.im - b.im); }
static inline cpx cmul(cpx a, cpx b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cpx cscale(cpx a, double s) { return cmk(a.re * s, a.im * s); }
static inline cpx cdivc(cpx a, cpx b) {
    double d = b.re * b.re + b.im * b.im;
    return cmk((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}
static inline cpx csqrt_(cpx z) {
    if (z.re == 0.0 && z.im == 0.0) return cmk(0, 0);
    double m = sqrt(z.re * z.re + z.im * z.im);
    double t = sqrt(0.5 * (m + fabs(z.re)));
    if (z.re >= 0.0) return cmk(t, z.im / (2.0 * t));
    return cmk(fabs(z.im) / (2.0 * t), z.im < 0.0 ? -t : t);
}
static inline cpx ccbrt_(cpx z) {
    if (z.re == 0.0 && z.im == 0.0) return cmk(0, 0);
    double m = cbrt(sqrt(z.re * z.re + z.im * z.im));
    double a = atan2(z.im, z.re) / 3.0;
    return cmk(m * cos(a), m * sin(a));
}

/*
 * The two functions below (solve_quartic, p3p) restate, operation by operation, the algorithm of Laurent Kneip's
 * solveQuartic.m / p3p.m as carried in the reference repository (old/solveQuartic.m, old/p3p.m), which are
 * distributed under the following licence; the notice is retained as that licence requires.
 *
 * Copyright (c) 2011, Laurent Kneip, ETH Zurich
 * All rights reserved.
 *
 * Redistribution and use in source and binary forms, with or without
 * modification, are permitted provided that the following conditions are met:
 *     * Redistributions of source code must retain the above copyright
 *       notice, this list of conditions and the following disclaimer.
 *     * Redistributions in binary form must reproduce the above copyright
 *       notice, this list of conditions and the following disclaimer in the
 *       documentation and/or other materials provided with the distribution.
 *     * Neither the name of ETH Zurich nor the
 *       names of its contributors may be used to endorse or promote products
 *       derived from this software without specific prior written permission.
 *
 * THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND
 * ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED
 * WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
 * DISCLAIMED. IN NO EVENT SHALL ETH ZURICH BE LIABLE FOR ANY
 * DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
 * (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES;
 * LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND
 * ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
 * (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS
 * SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
/* old/solveQuartic.m:26-66 */
static void solve_quartic(const double* f, cpx* roots) {
    const double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    const double A_pw2 = A * A, B_pw2 = B * B;
    const double A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B;
    const double A_pw4 = A_pw3 * A, B_pw4 = B_pw3 * B;
    const double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;                                             /* :41 */
    const double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;                             /* :42 */
    const double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A; /* :43 */
    const double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;
    const double P = -alpha_pw2 / 12 - gamma;                                                          /* :48 */
    const double Q = -alpha_pw3 / 108 + alpha * gamma / 3 - beta * beta / 8;                           /* :49 */
    const cpx R = cadd(cmk(-Q / 2, 0), csqrt_(cmk(Q * Q / 4 + P * P * P / 27, 0)));                    /* :50 */
    const cpx U = ccbrt_(R);                                                                           /* :51 */
    cpx y;
    if (U.re == 0.0 && U.im == 0.0) y = csub(cmk(-5 * alpha / 6, 0), ccbrt_(cmk(Q, 0)));               /* :53-54 */
    else y = cadd(csub(cmk(-5 * alpha / 6, 0), cdivc(cmk(P, 0), cscale(U, 3.0))), U);                  /* :56 */
    const cpx w = csqrt_(cadd(cmk(alpha, 0), cscale(y, 2.0)));                                         /* :59 */
    const cpx bw = cdivc(cmk(2 * beta, 0), w);
    const cpx base = cadd(cmk(3 * alpha, 0), cscale(y, 2.0));
    const cpx sp = csqrt_(cscale(cadd(base, bw), -1.0));
    const cpx sm = csqrt_(cscale(csub(base, bw), -1.0));
    const cpx off = cmk(-B / (4 * A), 0);
    roots[0] = cadd(off, cscale(cadd(w, sp), 0.5));                                                    /* :61 */
    roots[1] = cadd(off, cscale(csub(w, sp), 0.5));                                                    /* :62 */
    roots[2] = cadd(off, cscale(cadd(cscale(w, -1.0), sm), 0.5));                                      /* :63 */
    roots[3] = cadd(off, cscale(csub(cscale(w, -1.0), sm), 0.5));                                      /* :64 */
}

void vo_or_solve_quartic(const double* f, double* out) {
    cpx r[4];
    solve_quartic(f, r);
    for (int i = 0; i < 4; ++i) { out[2 * i] = r[i].re; out[2 * i + 1] = r[i].im; }
}

static inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void matvec3(const double* M, const double* x, double* y) { /* M row-major */
    for (int i = 0; i < 3; ++i) y[i] = dot3(M + 3 * i, x);
}

/* old/p3p.m:41-217.  Pw: 3 world points (row i = point i); fv: 3 unit bearings (row i). */
void vo_or_p3p(const double* Pw, const double* fv, double* poses) {
    memset(poses, 0, sizeof(double) * 48);                                   /* :45 */
    double P1[3], P2[3], P3[3], f1[3], f2[3], f3[3];
    memcpy(P1, Pw, 24); memcpy(P2, Pw + 3, 24); memcpy(P3, Pw + 6, 24);
    double v1[3], v2[3], cr[3];
    for (int i = 0; i < 3; ++i) { v1[i] = P2[i] - P1[i]; v2[i] = P3[i] - P1[i]; }
    cross3(v1, v2, cr);
    if (sqrt(dot3(cr, cr)) == 0.0) return;                                   /* :56-58 */
    memcpy(f1, fv, 24); memcpy(f2, fv + 3, 24); memcpy(f3, fv + 6, 24);
    double T[9], e3n;
    {
        double* e1 = T; double* e2 = T + 3; double* e3 = T + 6;
        memcpy(e1, f1, 24);
        cross3(f1, f2, e3);
        e3n = sqrt(dot3(e3, e3));
        for (int i = 0; i < 3; ++i) e3[i] /= e3n;
        cross3(e3, e1, e2);
    }
    double f3t[3];
    matvec3(T, f3, f3t);                                                     /* :74 */
    if (f3t[2] > 0) {                                                        /* :78-98 */
        memcpy(f1, fv + 3, 24); memcpy(f2, fv, 24); memcpy(f3, fv + 6, 24);
        double* e1 = T; double* e2 = T + 3; double* e3 = T + 6;
        memcpy(e1, f1, 24);
        cross3(f1, f2, e3);
        e3n = sqrt(dot3(e3, e3));
        for (int i = 0; i < 3; ++i) e3[i] /= e3n;
        cross3(e3, e1, e2);
        matvec3(T, f3, f3t);
        memcpy(P1, Pw + 3, 24); memcpy(P2, Pw, 24); memcpy(P3, Pw + 6, 24);
    }
    double N[9];
    {
        double* n1 = N; double* n2 = N + 3; double* n3 = N + 6;
        double d[3], nn;
        for (int i = 0; i < 3; ++i) n1[i] = P2[i] - P1[i];
        nn = sqrt(dot3(n1, n1));
        for (int i = 0; i < 3; ++i) n1[i] /= nn;
        for (int i = 0; i < 3; ++i) d[i] = P3[i] - P1[i];
        cross3(n1, d, n3);
        nn = sqrt(dot3(n3, n3));
        for (int i = 0; i < 3; ++i) n3[i] /= nn;
        cross3(n3, n1, n2);
    }
    double P3n[3], d31[3], d21[3];
    for (int i = 0; i < 3; ++i) { d31[i] = P3[i] - P1[i]; d21[i] = P2[i] - P1[i]; }
    matvec3(N, d31, P3n);                                                    /* :114 */
    const double d_12 = sqrt(dot3(d21, d21));                                /* :116 */
    const double f_1 = f3t[0] / f3t[2], f_2 = f3t[1] / f3t[2];               /* :117-118 */
    const double p_1 = P3n[0], p_2 = P3n[1];
    const double cos_beta = dot3(f1, f2);                                    /* :122 */
    double b = 1 / (1 - cos_beta * cos_beta) - 1;                            /* :123 */
    if (cos_beta < 0) b = -sqrt(b); else b = sqrt(b);                        /* :125-129 */

    const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
    const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
    const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;                                         /* :146-148 */
    fac[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12; /* :150-152 */
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
             f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 +
             2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
             2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;   /* :154-164 */
    fac[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b -
             2 * p_1 * p_2 * d_12_pw2 * b;                                                             /* :166-169 */
    fac[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 -
             p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
             p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;                        /* :171-179 */
    cpx x[4];
    solve_quartic(fac, x);                                                                             /* :183 */

    for (int i = 0; i < 4; ++i) {                                                                      /* :187-215 */
        const double xr = x[i].re;
        const double cot_alpha = (-f_1 * p_1 / f_2 - xr * p_2 + d_12 * b) / (-f_1 * xr * p_2 / f_2 + p_1 - d_12);
        const double cos_theta = xr;
        const double st2 = 1 - xr * xr;
        const double sin_theta = st2 >= 0 ? sqrt(st2) : 0.0; /* real() of an imaginary sqrt (ransacLocalization.m:64-65) */
        const double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
        double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
        if (cot_alpha < 0) cos_alpha = -cos_alpha;
        const double k = sin_alpha * b + cos_alpha;
        double C[3] = {d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k, sin_theta * d_12 * sin_alpha * k};
        double Cw[3];
        for (int r = 0; r < 3; ++r) /* C = P1 + N' * C  (:204) */
            Cw[r] = P1[r] + ((N[0 * 3 + r] * C[0] + N[1 * 3 + r] * C[1]) + N[2 * 3 + r] * C[2]);
        const double R[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                             sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                             0, -sin_theta, cos_theta};                                               /* :206-208 */
        /* R = N' * R' * T  (:210), evaluated (N'*R')*T */
        double NR[9], Rw[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) /* (N')[r][k] = N[k][r]; (R')[k][c] = R[c][k] */
                NR[r * 3 + c] = (N[0 * 3 + r] * R[c * 3 + 0] + N[1 * 3 + r] * R[c * 3 + 1]) + N[2 * 3 + r] * R[c * 3 + 2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rw[r * 3 + c] = (NR[r * 3 + 0] * T[0 * 3 + c] + NR[r * 3 + 1] * T[1 * 3 + c]) + NR[r * 3 + 2] * T[2 * 3 + c];
        memcpy(poses + 12 * i, Cw, 24);
        memcpy(poses + 12 * i + 3, Rw, 72);
    }
}

/* Philox4x32-10 */
static void philox4x32_10(uint32_t c[4], const uint32_t key[2]) {
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void vo_or_philox_block(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, 0, 0}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox4x32_10(c, key);
    memcpy(out, c, sizeof c);
}

void vo_or_philox_sample3(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
    uint32_t c[4] = {hyp, 0, 0, 0}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox4x32_10(c, key);
    uint32_t i0 = c[0] % m;
    uint32_t i1 = c[1] % (m - 1); if (i1 >= i0) ++i1;
    uint32_t i2 = c[2] % (m - 2);
    uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
    if (i2 >= lo) ++i2;
    if (i2 >= hi) ++i2;
    idx[0] = (int32_t)i0; idx[1] = (int32_t)i1; idx[2] = (int32_t)i2;
}

static inline void bearing(const double* K, double u, double v, double* f) {
    double y = (v - K[5]) / K[4];
    double x = ((u - K[2]) - K[1] * y) / K[0];
    double n = sqrt((x * x + y * y) + 1.0);
    f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
}

/* N9 scoring.  Each pose (R_cw row-major, t_cw) is folded once into the 3x4 projection
 * Pm = K [R_cw | t_cw]; the projection stays homogeneous, so the inlier test needs no division:
 * ez = (u*z - u~)^2 + (v*z - v~)^2 < thr^2 * z^2.  Explicit fused multiply-adds in a fixed order
 * (fma() is correctly rounded on every platform); only inliers pay for e2 = ez / z^2.  Equivalent to
 * errors = sum((p - proj).^2) < tol^2 of old/ransacLocalization.m:96-99 up to rounding at the
 * threshold. */
static void proj_matrix(const double* Rt, const double* K, double* P) {
    for (int c = 0; c < 4; ++c) {
        const double g0 = c < 3 ? Rt[c] : Rt[9], g1 = c < 3 ? Rt[3 + c] : Rt[10], g2 = c < 3 ? Rt[6 + c] : Rt[11];
        P[c] = fma(K[2], g2, fma(K[1], g1, K[0] * g0));
        P[4 + c] = fma(K[5], g2, K[4] * g1);
        P[8 + c] = g2;
    }
}
static inline void proj_terms(const double* P, const double* X, const double* xy, double* ez, double* z2) {
    const double up = fma(P[2], X[2], fma(P[1], X[1], fma(P[0], X[0], P[3])));
    const double vp = fma(P[6], X[2], fma(P[5], X[1], fma(P[4], X[0], P[7])));
    const double zc = fma(P[10], X[2], fma(P[9], X[1], fma(P[8], X[0], P[11])));
    const double du = fma(xy[0], zc, -up), dv = fma(xy[1], zc, -vp);
    *ez = fma(dv, dv, du * du);
    *z2 = zc * zc;
}

int vo_or_p3p_ransac(const double* xy, const double* X, int m, const double* K, const vo_or_ransac_opts* o,
                     const int32_t* sample_idx, double* Rout, double* tout, uint8_t* inliers, int* best_hyp,
                     int* best_sol, int* trials_used, int32_t* hyp_count, uint64_t* hyp_cost, double* hyp_poses) {
    if (inliers) memset(inliers, 0, (size_t)(m > 0 ? m : 0));
    if (best_hyp) *best_hyp = -1;
    if (best_sol) *best_sol = -1;
    if (trials_used) *trials_used = 0;
    for (int i = 0; i < 9; ++i) Rout[i] = (i % 4 == 0) ? 1.0 : 0.0;
    tout[0] = tout[1] = tout[2] = 0;
    if (m < 4) return 1;
    const int H = o->num_hyp;
    const double thr2 = o->max_reproj * o->max_reproj;
    const uint64_t qthr = (uint64_t)(thr2 * 1048576.0);
    int32_t* cnt = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)H);
    uint64_t* cost = (uint64_t*)malloc(sizeof(uint64_t) * 4 * (size_t)H);
    double* Rt = (double*)malloc(sizeof(double) * 48 * (size_t)H); /* per solution: R_cw(9), t_cw(3) */
    int nth = o->threads > 0 ? o->threads : 1;
    (void)nth;
#pragma omp parallel for num_threads(nth) schedule(dynamic, 8)
    for (int h = 0; h < H; ++h) {
        int32_t id[3];
        if (sample_idx) { id[0] = sample_idx[3 * h]; id[1] = sample_idx[3 * h + 1]; id[2] = sample_idx[3 * h + 2]; }
        else vo_or_philox_sample3(o->seed, (uint32_t)h, (uint32_t)m, id);
        double Pw[9], f[9], poses[48];
        for (int k = 0; k < 3; ++k) {
            memcpy(Pw + 3 * k, X + 3 * id[k], 24);
            bearing(K, xy[2 * id[k]], xy[2 * id[k] + 1], f + 3 * k);
        }
        vo_or_p3p(Pw, f, poses);
        for (int s = 0; s < 4; ++s) {
            const double* C = poses + 12 * s; const double* Rwc = C + 3;
            double* R = Rt + 48 * (size_t)h + 12 * s; double* t = R + 9;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rwc[c * 3 + r];          /* :66 */
            for (int r = 0; r < 3; ++r) t[r] = -((R[r * 3] * C[0] + R[r * 3 + 1] * C[1]) + R[r * 3 + 2] * C[2]); /* :67 */
            int c_ = 0; uint64_t q = 0;
            double Pm[12];
            proj_matrix(R, K, Pm);
            for (int i = 0; i < m; ++i) {
                double ez, z2;
                proj_terms(Pm, X + 3 * i, xy + 2 * i, &ez, &z2);
                if (ez < thr2 * z2) { ++c_; q += (uint64_t)((ez / z2) * 1048576.0); } else q += qthr;
            }
            cnt[4 * h + s] = c_; cost[4 * h + s] = q;
        }
    }
    /* sequential semantics of the RANSAC loop, ransacLocalization.m:44-130 */
    int bh = -1, bs = -1, bcnt = 0, used = 0; uint64_t bcost = UINT64_MAX;
    double N = (double)H;
    for (int h = 0; h < H; ++h) {
        int s_best = 0;
        for (int s = 1; s < 4; ++s) {
            if (o->mode == 0 ? (cnt[4 * h + s] > cnt[4 * h + s_best]) : (cost[4 * h + s] < cost[4 * h + s_best])) s_best = s;
        }
        int c_ = cnt[4 * h + s_best]; uint64_t q = cost[4 * h + s_best];
        int better = o->mode == 0 ? (c_ > bcnt && c_ >= o->min_inliers) : (q < bcost && c_ >= o->min_inliers);
        if (better) { bh = h; bs = s_best; bcnt = c_; bcost = q; }
        used = h + 1;
        if (o->confidence > 0) {
            double orat = 1.0 - (double)bcnt / (double)m;
            if (orat > 0.9) orat = 0.9;
            double wi = 1.0 - orat;
            N = log(1.0 - o->confidence) / log(1.0 - wi * wi * wi);   /* :121 */
            if (!(N > (double)(used + 1))) break;                        /* :44 `while num_iterations > i`, i = used + 1 */
        }
    }
    if (trials_used) *trials_used = used;
    if (hyp_count) memcpy(hyp_count, cnt, sizeof(int32_t) * 4 * (size_t)H);
    if (hyp_cost) memcpy(hyp_cost, cost, sizeof(uint64_t) * 4 * (size_t)H);
    if (hyp_poses) memcpy(hyp_poses, Rt, sizeof(double) * 48 * (size_t)H);
    int status = 2;
    if (bh >= 0) {
        const double* R = Rt + 48 * (size_t)bh + 12 * bs; const double* t = R + 9;
        memcpy(Rout, R, 72); memcpy(tout, t, 24);
        double Pm[12];
        proj_matrix(R, K, Pm);
        for (int i = 0; i < m; ++i) {
            double ez, z2;
            proj_terms(Pm, X + 3 * i, xy + 2 * i, &ez, &z2);
            inliers[i] = (uint8_t)(ez < thr2 * z2);
        }
        if (best_hyp) *best_hyp = bh;
        if (best_sol) *best_sol = bs;
        status = 0;
    }
    free(cnt); free(cost); free(Rt);
    return status;
}
