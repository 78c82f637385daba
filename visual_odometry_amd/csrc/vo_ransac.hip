// vo_ransac.hip -- P3P-RANSAC localisation.  Replaces the estimateWorldCameraPose call of the pose
// step (modules/VisualOdometry.m:285-289) with the open algorithm the reference carries in old/:
// Kneip P3P (old/p3p.m:41-217), Ferrari quartic with complex intermediates
// (old/solveQuartic.m:26-66) and the RANSAC loop of old/ransacLocalization.m:44-130.  Normative
// details (Philox sample stream, MSAC cost, adaptive stop): oracle/p3p.c header, N9.
//
//   k_hyp      one thread per hypothesis: sample 3 indices (seeded Philox4x32-10 or a caller table),
//              bearings, P3P -> 4 x (R_cw, t_cw)
//   k_score    one 256-thread block per 8 poses (SC_HB hypotheses, or 8 ids of the survivor list) and part of
//              the points: every thread walks its points once and scores the 8 poses per point from
//              LDS-broadcast pose rows; inlier counts and MSAC costs are integers, reduced across the
//              wavefront by shuffles, across the four waves through LDS and across blocks / launches by
//              atomic adds -- order independent, hence bit-exact
//   k_survivors  long runs: the poses that can still beat the first 256 hypotheses' record after a quarter
//              of the points; only they see the other three quarters
//   k_best     one wavefront: sequential RANSAC semantics (strict-improvement updates, adaptive
//              stop) evaluated as a chunked prefix scan
//   k_inliers  inlier mask of the winning pose
// All arithmetic is IEEE double in the oracle's operation order (-ffp-contract=off).
// Bound: FP64 VALU (two IEEE divisions per pose x point); points and poses are L2/LDS resident.
#include "vo_common.h"
#include "vo_dev_math.h"

namespace {

struct cpx { double re, im; };
__device__ __forceinline__ cpx cmk(double a, double b) { cpx z; z.re = a; z.im = b; return z; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return cmk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return cmk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cpx cscale(cpx a, double s) { return cmk(a.re * s, a.im * s); }
__device__ __forceinline__ cpx cdivc(cpx a, cpx b) {
    double d = b.re * b.re + b.im * b.im;
    return cmk((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}
__device__ __forceinline__ cpx csqrt_(cpx z) {
    if (z.re == 0.0 && z.im == 0.0) return cmk(0, 0);
    double m = sqrt(z.re * z.re + z.im * z.im);
    double t = sqrt(0.5 * (m + fabs(z.re)));
    if (z.re >= 0.0) return cmk(t, z.im / (2.0 * t));
    return cmk(fabs(z.im) / (2.0 * t), z.im < 0.0 ? -t : t);
}
__device__ __forceinline__ cpx ccbrt_(cpx z) {
    if (z.re == 0.0 && z.im == 0.0) return cmk(0, 0);
    double m = cbrt(sqrt(z.re * z.re + z.im * z.im));
    double a = atan2(z.im, z.re) / 3.0;
    return cmk(m * cos(a), m * sin(a));
}

// The two functions below (solve_quartic, p3p) restate, operation by operation, the algorithm of Laurent Kneip's
// solveQuartic.m / p3p.m as carried in the reference repository (old/solveQuartic.m, old/p3p.m), which are
// distributed under the following licence; the notice is retained as that licence requires.
//
// Copyright (c) 2011, Laurent Kneip, ETH Zurich
// All rights reserved.
//
// Redistribution and use in source and binary forms, with or without
// modification, are permitted provided that the following conditions are met:
//     * Redistributions of source code must retain the above copyright
//       notice, this list of conditions and the following disclaimer.
//     * Redistributions in binary form must reproduce the above copyright
//       notice, this list of conditions and the following disclaimer in the
//       documentation and/or other materials provided with the distribution.
//     * Neither the name of ETH Zurich nor the
//       names of its contributors may be used to endorse or promote products
//       derived from this software without specific prior written permission.
//
// THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND
// ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED
// WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
// DISCLAIMED. IN NO EVENT SHALL ETH ZURICH BE LIABLE FOR ANY
// DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
// (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES;
// LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND
// ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
// (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS
// SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
// old/solveQuartic.m:26-66
__device__ void solve_quartic(const double* f, cpx* roots) {
    const double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    const double A_pw2 = A * A, B_pw2 = B * B;
    const double A_pw3 = A_pw2 * A, B_pw3 = B_pw2 * B;
    const double A_pw4 = A_pw3 * A, B_pw4 = B_pw3 * B;
    const double alpha = -3 * B_pw2 / (8 * A_pw2) + C / A;
    const double beta = B_pw3 / (8 * A_pw3) - B * C / (2 * A_pw2) + D / A;
    const double gamma = -3 * B_pw4 / (256 * A_pw4) + B_pw2 * C / (16 * A_pw3) - B * D / (4 * A_pw2) + E / A;
    const double alpha_pw2 = alpha * alpha, alpha_pw3 = alpha_pw2 * alpha;
    const double P = -alpha_pw2 / 12 - gamma;
    const double Q = -alpha_pw3 / 108 + alpha * gamma / 3 - beta * beta / 8;
    const cpx R = cadd(cmk(-Q / 2, 0), csqrt_(cmk(Q * Q / 4 + P * P * P / 27, 0)));
    const cpx U = ccbrt_(R);
    cpx y;
    if (U.re == 0.0 && U.im == 0.0) y = csub(cmk(-5 * alpha / 6, 0), ccbrt_(cmk(Q, 0)));
    else y = cadd(csub(cmk(-5 * alpha / 6, 0), cdivc(cmk(P, 0), cscale(U, 3.0))), U);
    const cpx w = csqrt_(cadd(cmk(alpha, 0), cscale(y, 2.0)));
    const cpx bw = cdivc(cmk(2 * beta, 0), w);
    const cpx base = cadd(cmk(3 * alpha, 0), cscale(y, 2.0));
    const cpx sp = csqrt_(cscale(cadd(base, bw), -1.0));
    const cpx sm = csqrt_(cscale(csub(base, bw), -1.0));
    const cpx off = cmk(-B / (4 * A), 0);
    roots[0] = cadd(off, cscale(cadd(w, sp), 0.5));
    roots[1] = cadd(off, cscale(csub(w, sp), 0.5));
    roots[2] = cadd(off, cscale(cadd(cscale(w, -1.0), sm), 0.5));
    roots[3] = cadd(off, cscale(csub(cscale(w, -1.0), sm), 0.5));
}

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void matvec3(const double* M, const double* x, double* y) {
    for (int i = 0; i < 3; ++i) y[i] = dot3(M + 3 * i, x);
}
__device__ __forceinline__ void cp3(double* d, const double* s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }

// old/p3p.m:41-217.  Pw: 3 world points (row i = point i); fv: 3 unit bearings.  poses: 4 x 12
// [C(3) R(9 row-major, cam->world)]
struct P3PFront { double P1[3], N[9], T[9], f_1, f_2, p_1, p_2, d_12, b; double xr[4]; bool ok; };
// everything up to the quartic's roots (shared by the four solutions) ...
__device__ __forceinline__ void p3p_front(const double* Pw, const double* fv, P3PFront& F) {
    F.ok = false;
    double* const P1 = F.P1; double P2[3], P3[3], f1[3], f2[3], f3[3];
    cp3(P1, Pw); cp3(P2, Pw + 3); cp3(P3, Pw + 6);
    double v1[3], v2[3], cr[3];
    for (int i = 0; i < 3; ++i) { v1[i] = P2[i] - P1[i]; v2[i] = P3[i] - P1[i]; }
    cross3(v1, v2, cr);
    if (sqrt(dot3(cr, cr)) == 0.0) return;   // collinear points: no solutions (F.ok stays false)
    cp3(f1, fv); cp3(f2, fv + 3); cp3(f3, fv + 6);
    double* const T = F.T; double e3n;
    {
        double* e1 = T; double* e2 = T + 3; double* e3 = T + 6;
        cp3(e1, f1);
        cross3(f1, f2, e3);
        e3n = sqrt(dot3(e3, e3));
        for (int i = 0; i < 3; ++i) e3[i] /= e3n;
        cross3(e3, e1, e2);
    }
    double f3t[3];
    matvec3(T, f3, f3t);
    if (f3t[2] > 0) {
        cp3(f1, fv + 3); cp3(f2, fv); cp3(f3, fv + 6);
        double* e1 = T; double* e2 = T + 3; double* e3 = T + 6;
        cp3(e1, f1);
        cross3(f1, f2, e3);
        e3n = sqrt(dot3(e3, e3));
        for (int i = 0; i < 3; ++i) e3[i] /= e3n;
        cross3(e3, e1, e2);
        matvec3(T, f3, f3t);
        cp3(P1, Pw + 3); cp3(P2, Pw); cp3(P3, Pw + 6);
    }
    double* const N = F.N;
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
    matvec3(N, d31, P3n);
    const double d_12 = sqrt(dot3(d21, d21));
    const double f_1 = f3t[0] / f3t[2], f_2 = f3t[1] / f3t[2];
    const double p_1 = P3n[0], p_2 = P3n[1];
    const double cos_beta = dot3(f1, f2);
    double b = 1 / (1 - cos_beta * cos_beta) - 1;
    if (cos_beta < 0) b = -sqrt(b); else b = sqrt(b);

    const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
    const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
    const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
             f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 +
             2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
             2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
    fac[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b -
             2 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 -
             p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
             p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
    cpx x[4];
    solve_quartic(fac, x);
    for (int i = 0; i < 4; ++i) F.xr[i] = x[i].re;
    F.f_1 = f_1; F.f_2 = f_2; F.p_1 = p_1; F.p_2 = p_2; F.d_12 = d_12; F.b = b;
    F.ok = true;
}
// ... and the back-substitution of one root: pose12 = [C(3) R(9 row-major, cam->world)] (old/p3p.m:187-215)
__device__ __forceinline__ void p3p_root(const P3PFront& F, double xr, double* pose12) {
    const double f_1 = F.f_1, f_2 = F.f_2, p_1 = F.p_1, p_2 = F.p_2, d_12 = F.d_12, b = F.b;
    const double* const N = F.N; const double* const T = F.T; const double* const P1 = F.P1;
    {
        const double cot_alpha = (-f_1 * p_1 / f_2 - xr * p_2 + d_12 * b) / (-f_1 * xr * p_2 / f_2 + p_1 - d_12);
        const double cos_theta = xr;
        const double st2 = 1 - xr * xr;
        const double sin_theta = st2 >= 0 ? sqrt(st2) : 0.0;
        const double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
        double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
        if (cot_alpha < 0) cos_alpha = -cos_alpha;
        const double k = sin_alpha * b + cos_alpha;
        double C[3] = {d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k, sin_theta * d_12 * sin_alpha * k};
        double Cw[3];
        for (int r = 0; r < 3; ++r)
            Cw[r] = P1[r] + ((N[0 * 3 + r] * C[0] + N[1 * 3 + r] * C[1]) + N[2 * 3 + r] * C[2]);
        const double R[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                             sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                             0, -sin_theta, cos_theta};
        double NR[9], Rw[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                NR[r * 3 + c] = (N[0 * 3 + r] * R[c * 3 + 0] + N[1 * 3 + r] * R[c * 3 + 1]) + N[2 * 3 + r] * R[c * 3 + 2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rw[r * 3 + c] = (NR[r * 3 + 0] * T[0 * 3 + c] + NR[r * 3 + 1] * T[1 * 3 + c]) + NR[r * 3 + 2] * T[2 * 3 + c];
        cp3(pose12, Cw);
        for (int k2 = 0; k2 < 9; ++k2) pose12[3 + k2] = Rw[k2];
    }
}
__device__ void p3p_device(const double* Pw, const double* fv, double* poses) {
    for (int k = 0; k < 48; ++k) poses[k] = 0.0;
    P3PFront F;
    p3p_front(Pw, fv, F);
    if (!F.ok) return;
    for (int i = 0; i < 4; ++i) p3p_root(F, F.xr[i], poses + 12 * i);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], const uint32_t key[2]) {
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void philox_sample3(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
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

__device__ __forceinline__ void bearing(const double* K, double u, double v, double* f) {
    double y = (v - K[5]) / K[4];
    double x = ((u - K[2]) - K[1] * y) / K[0];
    double n = sqrt((x * x + y * y) + 1.0);
    f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
}

struct HypArgs {
    const double* xy; vo_strides sxy; const double* X; vo_strides sX; int M;
    const double* K; int H; uint64_t seed; const int32_t* sample_idx; double* Rt;
    int h0;                 // first hypothesis of this launch (a later stage of the adaptive run)
    const int32_t* stopped; // non-null: the earlier stage's "adaptive stop reached" flag; nothing to do when set
    const int32_t* M_dev;   // non-null: the point count lives on the device, M is only its upper bound
    // non-null: the run's sums start at zero HERE instead of in a launch of their own (k_zero): every hypothesis clears
    // its four counts and costs, hypothesis 0 the run's 256-byte block (bound, survivor counts, sieve marks)
    int32_t* zcnt; unsigned long long* zcost; unsigned long long* zblock;
};

// The pose chain may run on the ctx's lane stream beside a tracking launch that fills every SIMD (vo_ctx_lane_*):
// its waves then arbitrate for issue slots against older tracking waves and, being the younger ones, lose (k_hyp:
// 12 us alone, 240 us beside k_lk).  A raised wave priority lets these short kernels through.
#define VO_POSE_PRIO() __builtin_amdgcn_s_setprio(3)

// one hypothesis: sample, P3P, the four solutions as [R_cw | t_cw] into a.Rt (and into `lds`, 48 doubles, if given)
__device__ __forceinline__ void hyp_body(const HypArgs& a, int M, int h, double* lds) {
    double K[9];
    for (int k = 0; k < 9; ++k) K[k] = a.K[k];
    int32_t id[3];
    bool ok = true;
    if (a.sample_idx) {
        for (int k = 0; k < 3; ++k) { id[k] = a.sample_idx[3 * h + k]; if (id[k] < 0 || id[k] >= M) ok = false; }
    } else philox_sample3(a.seed, (uint32_t)h, (uint32_t)M, id);
    double poses[48];
    if (ok) {
        double Pw[9], f[9];
        for (int k = 0; k < 3; ++k) {
            const long i = id[k];
            for (int c = 0; c < 3; ++c) Pw[3 * k + c] = a.X[i * a.sX.xi + c * a.sX.xc];
            bearing(K, a.xy[i * a.sxy.xi], a.xy[i * a.sxy.xi + a.sxy.xc], f + 3 * k);
        }
        p3p_device(Pw, f, poses);
    } else {
        for (int k = 0; k < 48; ++k) poses[k] = 0.0; // out-of-range table entry: a null hypothesis
    }
    for (int s = 0; s < 4; ++s) {
        const double* C = poses + 12 * s; const double* Rwc = C + 3;
        double R[9], t[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rwc[c * 3 + r];
        for (int r = 0; r < 3; ++r) t[r] = -((R[r * 3] * C[0] + R[r * 3 + 1] * C[1]) + R[r * 3 + 2] * C[2]);
        double* o = a.Rt + 48 * (long)h + 12 * s;
        for (int k = 0; k < 9; ++k) o[k] = R[k];
        for (int k = 0; k < 3; ++k) o[9 + k] = t[k];
        if (lds) {
            for (int k = 0; k < 9; ++k) lds[12 * s + k] = R[k];
            for (int k = 0; k < 3; ++k) lds[12 * s + 9 + k] = t[k];
        }
    }
}

// the same on FOUR lanes, one per solution (k_pose_small): sampling and everything up to the quartic's roots are
// computed by all four, the back-substitution and the change to [R_cw | t_cw] -- more than half of a thread's
// instructions -- by the lane of the root.  A wave alone on its SIMD is bound by its instruction count.
__device__ __forceinline__ void hyp_body4(const HypArgs& a, int M, int h, int r, double* lds) {
    double K[9];
    for (int k = 0; k < 9; ++k) K[k] = a.K[k];
    int32_t id[3];
    bool ok = true;
    if (a.sample_idx) {
        for (int k = 0; k < 3; ++k) { id[k] = a.sample_idx[3 * h + k]; if (id[k] < 0 || id[k] >= M) ok = false; }
    } else philox_sample3(a.seed, (uint32_t)h, (uint32_t)M, id);
    double pose[12];
    for (int k = 0; k < 12; ++k) pose[k] = 0.0;   // a null hypothesis (table entry out of range, collinear points)
    if (ok) {
        double Pw[9], f[9];
        for (int k = 0; k < 3; ++k) {
            const long i = id[k];
            for (int c = 0; c < 3; ++c) Pw[3 * k + c] = a.X[i * a.sX.xi + c * a.sX.xc];
            bearing(K, a.xy[i * a.sxy.xi], a.xy[i * a.sxy.xi + a.sxy.xc], f + 3 * k);
        }
        P3PFront F;
        p3p_front(Pw, f, F);
        if (F.ok) p3p_root(F, r == 0 ? F.xr[0] : (r == 1 ? F.xr[1] : (r == 2 ? F.xr[2] : F.xr[3])), pose);
    }
    const double* C = pose; const double* Rwc = pose + 3;
    double R[9], t[3];
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c) R[i * 3 + c] = Rwc[c * 3 + i];
    for (int i = 0; i < 3; ++i) t[i] = -((R[i * 3] * C[0] + R[i * 3 + 1] * C[1]) + R[i * 3 + 2] * C[2]);
    double* o = a.Rt + 48 * (long)h + 12 * r;
    for (int k = 0; k < 9; ++k) { o[k] = R[k]; lds[12 * r + k] = R[k]; }
    for (int k = 0; k < 3; ++k) { o[9 + k] = t[k]; lds[12 * r + 9 + k] = t[k]; }
}

__global__ __launch_bounds__(64, 4) void k_hyp(HypArgs a) {   // <= 128 VGPRs: fits the hole one finished tracking wave leaves (160 did not: the kernel starved beside k_lk)
    VO_POSE_PRIO();
    const int h = a.h0 + blockIdx.x * 64 + threadIdx.x;
    if (h >= a.H || (a.stopped && *a.stopped)) return;
    if (a.zcnt) {
        *reinterpret_cast<int4*>(a.zcnt + 4 * (long)h) = make_int4(0, 0, 0, 0);
        *reinterpret_cast<ulonglong2*>(a.zcost + 4 * (long)h) = make_ulonglong2(0ull, 0ull);
        *reinterpret_cast<ulonglong2*>(a.zcost + 4 * (long)h + 2) = make_ulonglong2(0ull, 0ull);
        if (h == 0)
            for (int k = 0; k < 32; ++k) a.zblock[k] = 0ull;
    }
    if (a.M_dev) { a.M = min(a.M, *a.M_dev); if (a.M < 4) return; }
    hyp_body(a, a.M, h, nullptr);
}

constexpr int SC_HB = 2;          // hypotheses per block
constexpr int SC_NS = 4 * SC_HB;  // poses per block
constexpr int SC_PT = 4;          // points per thread and pass

struct ScoreArgs {
    const double* xy; vo_strides sxy; const double* X; vo_strides sX; int M;
    const double* K; int H; double thr2; const double* Rt; int32_t* cnt; unsigned long long* cost;
    unsigned long long* counters;
    int h0;                 // contiguous form: first hypothesis of this launch (block b scores hypotheses h0 + 2b, h0 + 2b + 1)
    const int32_t* list;    // list form (non-null): pose ids 4h + s, eight per block, *list_n of them
    const int32_t* list_n;
    int q0, q1;             // the launch scores points [M Q0 / 64, M Q1 / 64); blockIdx.y takes one of gridDim.y equal parts.
                            // qdev null: Q = q.  qdev non-null: q indexes {0, qdev[0], qdev[1], 64} (sieve_mark)
    const int32_t* qdev;    // the two sieve marks of a staged run, chosen on the device by the first stage's k_best
    const int32_t* stopped; // as in HypArgs
    const int32_t* M_dev;   // as in HypArgs
    unsigned long long* tests;   // optional: one word that counts the pose x point tests executed (bench.py's k_score roofline)
};

// Scores 8 poses per block on a range of the points and ADDS what it finds to cnt / cost (zeroed by k_zero):
// the inlier count, and the MSAC cost of the range = inlier terms + (points of the range - inliers) * qthr.  All
// sums are integers, so any split of the points over blocks and launches gives the same totals.
// One wave per block, under 1 KB of LDS and 128 VGPRs: beside a tracking launch that fills the chip with one-wave
// workgroups (vo_ctx_lane_*), such a block fits the hole any finished tracking wave leaves; a 256-thread block had to
// wait for four holes on one CU at once and starved (as did the runtime's memset kernel: hence k_zero).
constexpr int SC_NT = 64;
__device__ __forceinline__ int sieve_mark(const int32_t* qdev, int q) {
    if (!qdev) return q;
    return q <= 0 ? 0 : (q == 1 ? qdev[0] : (q == 2 ? qdev[1] : 64));
}
// the body of k_score.  own_poses (k_pose_small): the block's eight poses lie in LDS (the block generated them itself)
// and the block sees every point, so its sums are stored, not added
__device__ __forceinline__ void score_body(ScoreArgs a, const double* own_poses) {
    __shared__ double sRt[SC_NS * 12];
    __shared__ int s_pid[SC_NS];
    __shared__ int s_cnt[SC_NS];
    __shared__ unsigned long long s_cost[SC_NS];
    if (a.M_dev) { a.M = min(a.M, *a.M_dev); if (a.M < 4) return; }
    const int tid = threadIdx.x;
    if (tid < SC_NS) { s_cnt[tid] = 0; s_cost[tid] = 0ull; }   // (ordered before their use by the barriers below)
    const int nlist = a.list ? *a.list_n : 0;
    if (a.list && (int)blockIdx.x * SC_NS >= nlist) return;   // uniform
    if (tid < SC_NS) {
        int pid;
        if (a.list) { const int k = blockIdx.x * SC_NS + tid; pid = k < nlist ? a.list[k] : -1; }
        else { pid = 4 * a.h0 + (int)blockIdx.x * SC_NS + tid; if (pid >= 4 * a.H) pid = -1; }
        s_pid[tid] = pid;
    }
    __syncthreads();
    for (int k = tid; k < SC_NS * 12; k += SC_NT) {
        const int hs = k / 12;               // pose within the block
        const int e = k - 12 * hs;           // element of the pose's 3x4 projection
        const int pid = s_pid[hs];
        const double* Rt = own_poses ? own_poses + 12 * hs : a.Rt + 12 * (long)pid;
        sRt[k] = pid >= 0 ? proj_elem(Rt, a.K, e >> 2, e & 3) : 0.0;
    }
    __syncthreads();
    const long r0 = (long)a.M * sieve_mark(a.qdev, a.q0) / 64, r1 = (long)a.M * sieve_mark(a.qdev, a.q1) / 64;
    const long per = (r1 - r0 + gridDim.y - 1) / gridDim.y;
    const int c0 = (int)(r0 + per * blockIdx.y), c1 = (int)min(r0 + per * (blockIdx.y + 1), r1);
    const double thr2 = a.thr2;
    const unsigned long long qthr = (unsigned long long)(thr2 * 1048576.0);
    int cnt[SC_NS];
    unsigned long long cost[SC_NS]; // inlier terms only; the outliers' (points - cnt) * qthr is added at the end
#pragma unroll
    for (int s = 0; s < SC_NS; ++s) { cnt[s] = 0; cost[s] = 0; }
    // SC_PT points per thread and pass: a pose read from LDS serves all of them
    for (int i0 = c0 + tid; i0 < c1; i0 += SC_NT * SC_PT) {
        double X0[SC_PT], X1[SC_PT], X2[SC_PT], u[SC_PT], v[SC_PT];
        bool has[SC_PT];
#pragma unroll
        for (int p = 0; p < SC_PT; ++p) {
            const int i = i0 + SC_NT * p;
            has[p] = i < c1;
            const int j = has[p] ? i : i0;
            X0[p] = a.X[j * a.sX.xi]; X1[p] = a.X[j * a.sX.xi + a.sX.xc]; X2[p] = a.X[j * a.sX.xi + 2 * a.sX.xc];
            u[p] = a.xy[j * a.sxy.xi]; v[p] = a.xy[j * a.sxy.xi + a.sxy.xc];
        }
#pragma unroll
        for (int s = 0; s < SC_NS; ++s) {
            double P[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = sRt[12 * s + k];
            ProjT t[SC_PT];
            bool in[SC_PT], any = false;
#pragma unroll
            for (int p = 0; p < SC_PT; ++p) {
                t[p] = proj_terms(P, X0[p], X1[p], X2[p], u[p], v[p]);
                in[p] = has[p] && (t[p].ez < thr2 * t[p].z2);
                cnt[s] += (int)in[p];
                any = any || in[p];
            }
            if (any) { // a real branch (the empty asm keeps it from being if-converted): most waves of
                       // most poses hold no inlier and skip the division
                asm volatile("" ::: "memory");
                unsigned long long q = 0;
#pragma unroll
                for (int p = 0; p < SC_PT; ++p)
                    if (in[p]) q += (unsigned long long)((t[p].ez / t[p].z2) * 1048576.0);
                cost[s] += q;
            }
        }
    }
    if (c1 <= c0) return;
    // The wave's sums per pose (integer sums: any order gives the same bits).  Most poses of most blocks have no inlier at
    // all -- one compare and a branch; a pose with a few (a wrong pose that happens to pass near some points) adds them
    // with LDS atomics; only a pose with inliers on many lanes -- a good one, one in eight -- pays the butterfly (18
    // ds_bpermute and as many adds), which every pose paid before: half of the kernel's vector instructions in the short
    // pieces the sieve's later stages are made of (SQ_INSTS_VALU per wave: 1024, of which the tests ~520).  (LDS atomics
    // for every pose were measured too: 64 lanes on one address serialise -- 0.27 instead of 0.19 ms for the scoring.)
#pragma unroll
    for (int s = 0; s < SC_NS; ++s) {
        const unsigned long long bal = __ballot(cnt[s] != 0);
        if (bal == 0ull) continue;
        if (__popcll(bal) <= 4) {
            if (cnt[s]) { atomicAdd(&s_cnt[s], cnt[s]); atomicAdd(&s_cost[s], cost[s]); }
        } else {
            int c = cnt[s]; unsigned long long q = cost[s];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { c += __shfl_xor(c, d, 64); q += __shfl_xor(q, d, 64); }
            if (tid == 0) { s_cnt[s] = c; s_cost[s] = q; }
        }
    }
    __syncthreads();
    if (tid < SC_NS && s_pid[tid] >= 0) {
        const int c = s_cnt[tid];
        const unsigned long long total = s_cost[tid] + (unsigned long long)((c1 - c0) - c) * qthr;
        if (own_poses) { a.cnt[s_pid[tid]] = c; a.cost[s_pid[tid]] = total; }
        else {
            if (c) atomicAdd(&a.cnt[s_pid[tid]], c);
            atomicAdd(&a.cost[s_pid[tid]], total);
        }
    }
    if (tid == 0 && a.counters && !a.list && blockIdx.y == 0)
        atomicAdd(&a.counters[4 * (blockIdx.x & (VO_COUNTER_SHARDS - 1)) + 3], (unsigned long long)min(SC_HB, a.H - (a.h0 + (int)blockIdx.x * SC_HB)));
    if (tid == 0 && a.tests) {
        int live = 0;
        for (int s = 0; s < SC_NS; ++s) live += s_pid[s] >= 0;
        atomicAdd(a.tests, (unsigned long long)live * (unsigned long long)(c1 - c0));
    }
}
__global__ __launch_bounds__(SC_NT) void k_score(ScoreArgs a) {
    VO_POSE_PRIO();
    if (a.stopped && *a.stopped) return;   // uniform: the whole launch has nothing to do
    score_body(a, nullptr);
}

// zeroes the run's sums (counts, costs, the bound and survivor-count words): one-wave blocks, see k_score
__global__ __launch_bounds__(64) void k_zero(unsigned* p, long nwords) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}

// Which poses of the hypotheses [h0, H) can still become the record after the points up to sieve mark `seen`?
// bound[0] = the cost (mode 1) or the inlier count (mode 0) of the best record among the hypotheses BEFORE h0, written
// by the first stage's k_best.  After n points a pose's final cost is at least its cost so far, its final count at most
// its count so far + (M - n).  A later hypothesis replaces the record only by strict improvement
// (old/ransacLocalization.m:107), and every hypothesis behind the bound is earlier in the order, so a pose that
// cannot strictly improve on the bound could never have become the record: it is left out of the remaining points.
// Record, trial count, pose and inlier set are those of the full scoring; only the per-hypothesis counts / costs of
// the poses left out stay partial sums (callers who read them get the full scoring: score_all).
// Candidates: every pose of [h0, H) (in == nullptr) or the previous round's list.
struct SurvArgs {
    const int32_t* cnt; const unsigned long long* cost; const unsigned long long* bound; int h0, H, M, mode, seen;
    const int32_t* in; const int32_t* in_n; int32_t* list; int32_t* list_n; const int32_t* stopped; const int32_t* M_dev;
    const int32_t* qdev;    // as in ScoreArgs: `seen` indexes the sieve marks
};
__global__ __launch_bounds__(64) void k_survivors(SurvArgs a) {
    VO_POSE_PRIO();
    if (a.stopped && *a.stopped) return;
    if (a.M_dev) { a.M = min(a.M, *a.M_dev); if (a.M < 4) return; }
    const int k = blockIdx.x * 64 + threadIdx.x;
    int pid;
    if (a.in) { if (k >= *a.in_n) return; pid = a.in[k]; }
    else { pid = 4 * a.h0 + k; if (pid >= 4 * a.H) return; }
    const unsigned long long bound = a.bound[0];
    const long seen = (long)a.M * sieve_mark(a.qdev, a.seen) / 64;      // where k_score's ranges so far end
    const bool alive = a.mode == 1 ? (a.cost[pid] < bound)
                                   : ((unsigned long long)((long)a.cnt[pid] + ((long)a.M - seen)) > bound);
    if (alive) a.list[atomicAdd(a.list_n, 1)] = pid;
}

// ---- sequential RANSAC semantics (old/ransacLocalization.m:86-124) as a chunked scan, one block ----
struct BState { int h, s, c; unsigned long long q; };

__device__ __forceinline__ bool improves(int mode, int min_inl, int c, unsigned long long q, const BState& st) {
    if (c < min_inl) return false;
    return mode == 0 ? (c > st.c) : (q < st.q);
}
__device__ __forceinline__ void hyp_best(const int32_t* cnt, const unsigned long long* cost, int mode, int h, int& s_best,
                                         int& c, unsigned long long& q) {
    // the four solutions' records in two vector loads each, compared in registers (a compare that re-reads
    // cnt[4h + s_best] chains up to three dependent global loads per hypothesis)
    const int4 c4 = *reinterpret_cast<const int4*>(cnt + 4 * (long)h);
    const ulonglong2 qa = *reinterpret_cast<const ulonglong2*>(cost + 4 * (long)h);
    const ulonglong2 qb = *reinterpret_cast<const ulonglong2*>(cost + 4 * (long)h + 2);
    const int cs[4] = {c4.x, c4.y, c4.z, c4.w};
    const unsigned long long qs[4] = {qa.x, qa.y, qb.x, qb.y};
    s_best = 0; c = cs[0]; q = qs[0];
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (mode == 0 ? (cs[s] > c) : (qs[s] < q)) { s_best = s; c = cs[s]; q = qs[s]; }
}
__device__ __forceinline__ double ransac_limit(double confidence, int bcnt, int m, int sample_size) {
    double orat = 1.0 - (double)bcnt / (double)m;
    if (orat > 0.9) orat = 0.9;
    double wi = 1.0 - orat;
    double wk;
    if (sample_size == 8) { const double w2 = wi * wi, w4 = w2 * w2; wk = w4 * w4; } // essential.c
    else if (sample_size == 5) { const double w2 = wi * wi, w4 = w2 * w2; wk = w4 * wi; } // five_point.c
    else wk = wi * wi * wi;                                                            // old/ransacLocalization.m:121
    return log(1.0 - confidence) / log(1.0 - wk);
}

struct BestArgs {
    const int32_t* cnt; const unsigned long long* cost; const double* Rt; int H, M, mode, min_inl;
    double confidence; double* out; // R_cw(9) t_cw(3) then 6 int32: status, hyp, sol, trials, inliers, stopped
    int sample_size;                // 3 (P3P), 8 (eight-point) or 5 (five-point)
    int rows_per_trial = 1;         // five-point: a sample's ten solutions are ten rows of the table; the stop is tested once per sample
    int later_stage;                // 1: an earlier stage ran on a prefix; keep its record if it reached the stop
    const int32_t* M_dev;           // as in HypArgs; fewer than 4 points: status 1 and the identity
    unsigned long long* bound_out;  // non-null (first stage): the record's cost / count for k_score's early out
    int32_t* sieve_out;             // with bound_out: the two sieve marks (64ths of the points), see below
    unsigned long long qthr;        // an outlier's cost term in the fixed point of `cost`
};

__device__ __forceinline__ BState combine(int mode, int min_inl, const BState& earlier, const BState& later) {
    // state after an earlier span followed by a later span: the later record wins only if it strictly improves
    const bool later_wins = later.h >= 0 && (earlier.h < 0 || improves(mode, min_inl, later.c, later.q, earlier));
    return later_wins ? later : earlier;
}
__device__ __forceinline__ BState shfl_up_state(const BState& x, int d) {
    BState o;
    o.h = __shfl_up(x.h, d, 64); o.s = __shfl_up(x.s, d, 64); o.c = __shfl_up(x.c, d, 64); o.q = __shfl_up(x.q, d, 64);
    return o;
}

// One block of BEST_NT threads, each owning a contiguous chunk of hypotheses: local pass, two-level
// scan of the best-so-far records (wave shuffles, then the 16 wave totals through LDS), then a
// re-walk of the chunk with the carried-in record that looks for the adaptive stop.
// BEST_NT = 256 alone on the chip (1024 and 64 were measured too: 8 and 6 us slower per 2000-hypothesis launch);
// BEST_NT = 64 beside a tracking launch (lane): a four-wave workgroup waits ~180 us for a CU with four free wave
// slots on four SIMDs at once, a one-wave workgroup takes the first hole.
constexpr int RANSAC_STAGE = 256;  // hypotheses of the first stage of an adaptive run (a multiple of 64 and of SC_HB)
template <int BEST_NT>
__device__ __forceinline__ void best_body(BestArgs a) {
    __shared__ BState s_wave[BEST_NT / 64];
    __shared__ int s_stop;
    __shared__ BState s_fin;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (a.later_stage && ((const int32_t*)(a.out + 12))[5]) return;   // uniform
    if (a.M_dev) {
        a.M = min(a.M, *a.M_dev);
        if (a.M < 4) {                                                 // estimateWorldCameraPose needs 4 points
            if (tid == 0) {
                for (int k = 0; k < 9; ++k) a.out[k] = (k % 4 == 0) ? 1.0 : 0.0;
                a.out[9] = a.out[10] = a.out[11] = 0.0;
                int32_t* info = (int32_t*)(a.out + 12);
                info[0] = 1; info[1] = -1; info[2] = -1; info[3] = 0; info[4] = 0; info[5] = 1;
            }
            return;
        }
    }
    const int per = (a.H + BEST_NT - 1) / BEST_NT;
    const int lo = min(tid * per, a.H), hi = min(lo + per, a.H);
    BState none; none.h = -1; none.s = -1; none.c = 0; none.q = ~0ull;
    if (tid == 0) s_stop = 0x7fffffff;
    BState st = none;
    for (int h = lo; h < hi; ++h) {
        int s, c; unsigned long long q;
        hyp_best(a.cnt, a.cost, a.mode, h, s, c, q);
        if (improves(a.mode, a.min_inl, c, q, st)) { st.h = h; st.s = s; st.c = c; st.q = q; }
    }
    BState inc = st; // inclusive scan over the lanes of the wave
    for (int d = 1; d < 64; d <<= 1) {
        const BState o = shfl_up_state(inc, d);
        if (lane >= d) inc = combine(a.mode, a.min_inl, o, inc);
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    BState wcarry = none;
    for (int w = 0; w < wid; ++w) {
        BState sw;   // member-wise out of LDS: a reference to either LDS or registers makes these flat loads
        sw.h = s_wave[w].h; sw.s = s_wave[w].s; sw.c = s_wave[w].c; sw.q = s_wave[w].q;
        wcarry = combine(a.mode, a.min_inl, wcarry, sw);
    }
    BState lcarry = shfl_up_state(inc, 1);
    if (lane == 0) lcarry = none;
    const BState carry = combine(a.mode, a.min_inl, wcarry, lcarry);
    // re-walk the chunk with the carried-in record, looking for the adaptive stop
    BState cur = carry;
    int stop = 0x7fffffff;
    if (a.confidence > 0) {
        double N = ransac_limit(a.confidence, cur.c, a.M, a.sample_size);
        for (int h = lo; h < hi; ++h) {
            int s, c; unsigned long long q;
            hyp_best(a.cnt, a.cost, a.mode, h, s, c, q);
            if (improves(a.mode, a.min_inl, c, q, cur)) {
                cur.h = h; cur.s = s; cur.c = c; cur.q = q;
                N = ransac_limit(a.confidence, cur.c, a.M, a.sample_size);
            }
            if (a.rows_per_trial == 1) {
                if (!(N > (double)(h + 2))) { stop = h; break; } // :44 `while num_iterations > i`
            } else if (h % a.rows_per_trial == a.rows_per_trial - 1 && !(N > (double)(h / a.rows_per_trial + 2))) { stop = h; break; }
        }
        if (stop != 0x7fffffff) atomicMin(&s_stop, stop);
    }
    __syncthreads();
    const int gstop = s_stop;
    // the record at the stop belongs to the thread whose chunk holds it; without a stop it is the
    // record after every chunk, which the last thread knows
    const bool owner = gstop == 0x7fffffff ? (tid == BEST_NT - 1) : (stop == gstop);
    if (owner) s_fin = gstop == 0x7fffffff ? combine(a.mode, a.min_inl, wcarry, inc) : cur;
    __syncthreads();
    if (tid == 0) {
        const BState fin = s_fin;
        int32_t* info = (int32_t*)(a.out + 12);
        const int used = (gstop == 0x7fffffff ? a.H : gstop + 1) / a.rows_per_trial;
        if (fin.h >= 0) {
            const double* p = a.Rt + 48 * (long)fin.h + 12 * fin.s;
            for (int k = 0; k < 12; ++k) a.out[k] = p[k];
            info[0] = 0;
        } else {
            for (int k = 0; k < 9; ++k) a.out[k] = (k % 4 == 0) ? 1.0 : 0.0;
            a.out[9] = a.out[10] = a.out[11] = 0.0;
            info[0] = 2;
        }
        info[1] = fin.h; info[2] = fin.s; info[3] = used; info[4] = 0;
        if (a.bound_out) {  // no record yet: a bound no pose can fail
            a.bound_out[0] = fin.h >= 0 ? (a.mode == 1 ? fin.q : (unsigned long long)fin.c) : (a.mode == 1 ? ~0ull : 0ull);
            // Where to sieve.  A pose without inliers falls behind the record once rho = (record's cost) / (cost of
            // M outliers) of the points are seen (count mode: rho = 1 - record's count / M); most of the second
            // stage's poses are of that kind, so the first mark sits just behind rho -- every 64th of the points
            // before it is work on poses that cannot win yet cannot be dropped, every 64th behind it is work on poses
            // that could have been dropped.  The second mark halves what is left.  Any marks give the same record
            // (k_survivors' argument does not depend on them); these only decide how much is scored.
            double rho = 1.0;
            if (fin.h >= 0)
                rho = a.mode == 1 ? (double)fin.q / ((double)a.M * (double)a.qthr) : 1.0 - (double)fin.c / (double)a.M;
            int f1 = (int)(rho * 68.0) + 2;          // 64 rho, 6 % further, rounded up and one more
            f1 = f1 < 4 ? 4 : (f1 > 56 ? 56 : f1);
            const int rest = (64 - f1) / 2;
            a.sieve_out[0] = f1;
            a.sieve_out[1] = f1 + (rest < 2 ? 2 : rest);
        }
        info[5] = gstop == 0x7fffffff ? 0 : 1;   // the adaptive stop was reached inside these H hypotheses
    }
}
template <int BEST_NT>
__global__ __launch_bounds__(BEST_NT) void k_best(BestArgs a) {
    VO_POSE_PRIO();
    best_body<BEST_NT>(a);
}

// ---- short point sets (the frame loop: 300 landmarks): a stage of the adaptive run in ONE launch ----
// k_hyp -> k_score -> k_best are three launches of a few microseconds of work each when the points are few, and two
// stages make six.  Here a one-wave block generates its two hypotheses (four lanes each, one per solution of the
// P3P solve), scores their eight poses on ALL points (so its sums are stores, nothing has to be zeroed), and
// the block that finishes last -- a ticket counted up behind a fence -- runs k_best's scan.  Same device functions,
// same bits (test_ransac_adaptive_run_in_two_stages runs both forms against the oracle).
struct SmallArgs { HypArgs h; ScoreArgs s; BestArgs b; int32_t* ticket; int nblocks; };
constexpr int RANSAC_SMALL_M = 1024;   // points up to which a stage is one launch

__global__ __launch_bounds__(64, 4) void k_pose_small(SmallArgs a) {
    VO_POSE_PRIO();
    __shared__ double s_pose[SC_HB * 48];
    __shared__ int s_last;
    if (a.h.stopped && *a.h.stopped) return;             // a later stage behind a stop that was reached: uniform
    int M = a.h.M;
    if (a.h.M_dev) M = min(M, *a.h.M_dev);
    if (M < 4) { if (blockIdx.x == 0) best_body<64>(a.b); return; }   // status 1 and the identity (uniform per block)
    const int tid = threadIdx.x;
    if (tid < 4 * SC_HB) {                                // four lanes per hypothesis, one per solution
        const int h = a.h.h0 + (int)blockIdx.x * SC_HB + (tid >> 2);
        if (h < a.h.H) hyp_body4(a.h, M, h, tid & 3, s_pose + 48 * (tid >> 2));
    }
    __syncthreads();
    score_body(a.s, s_pose);
    __threadfence();                                     // this block's poses and sums are out before its ticket
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.ticket, 1) == a.nblocks - 1;
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) *a.ticket = 0;                         // for the next launch (behind this one in its stream)
    __threadfence();
    best_body<64>(a.b);
}

struct InlArgs {
    const double* xy; vo_strides sxy; const double* X; vo_strides sX; int M; const double* K; double thr2;
    double* out; uint8_t* inliers;
    const int32_t* M_dev;   // as in HypArgs (flags beyond the device count are written as 0)
    double* pose_out;       // optional: the pose [R_WC; t_WC] of the record (modules/VisualOdometry.m:318), C = -R_cw^T t_cw
};
__global__ __launch_bounds__(64) void k_inliers(InlArgs a) {
    VO_POSE_PRIO();
    const int i = blockIdx.x * 64 + threadIdx.x;
    int32_t* info = (int32_t*)(a.out + 12);
    if (i == 0 && a.pose_out) {
        const double* out = a.out;
        for (int k = 0; k < 9; ++k) a.pose_out[k] = out[k];
        for (int c = 0; c < 3; ++c) a.pose_out[9 + c] = -((out[c] * out[9] + out[3 + c] * out[10]) + out[6 + c] * out[11]);
    }
    bool in = false;
    const int Md = a.M_dev ? min(a.M, *a.M_dev) : a.M;
    if (i < Md && info[0] == 0) {
        double P[12];
        for (int k = 0; k < 12; ++k) P[k] = proj_elem(a.out, a.K, k >> 2, k & 3);
        const ProjT p = proj_terms(P, a.X[i * a.sX.xi],
                                   a.X[i * a.sX.xi + a.sX.xc], a.X[i * a.sX.xi + 2 * a.sX.xc], a.xy[i * a.sxy.xi],
                                   a.xy[i * a.sxy.xi + a.sxy.xc]);
        in = p.ez < a.thr2 * p.z2;
    }
    if (i < a.M) a.inliers[i] = in ? 1 : 0;
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&info[4], __popcll(bal));
}

__global__ __launch_bounds__(64) void k_p3p_batch(const double* P, const double* f, int n, double* poses) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double Pw[9], fv[9], out[48];
    for (int k = 0; k < 9; ++k) { Pw[k] = P[9 * (long)i + k]; fv[k] = f[9 * (long)i + k]; }
    p3p_device(Pw, fv, out);
    for (int k = 0; k < 48; ++k) poses[48 * (long)i + k] = out[k];
}

} // namespace

static int score_chunks(int nblocks, int npoints) {
    // one-wave blocks: split the points over blockIdx.y until the launch has about a wave per SIMD (1024 blocks; 3072
    // -- 4 waves per SIMD, the first choice -- is 0.01 ms faster for the chain alone and 0.005-0.01 ms slower for the
    // step and the frame loop at the stress size, where the launch competes with tracking waves for slots; a piece no
    // shorter than 128 points; pieces of at least 1024 points until round 3: the frame loop at 2k landmarks runs its
    // 64-hypothesis prefix over 2000 points, a launch of 16 waves each walking all of them -- 0.240 -> 0.222 ms per
    // frame with the short pieces, no difference at 25k points.  Counts and costs are integer sums: the same bits)
    static const int target = [] { const char* e = getenv("VO_SCORE_BLOCKS"); return e ? atoi(e) : 1024; }();   // A/B switch
    static const int piece = [] { const char* e = getenv("VO_SCORE_PIECE"); return e ? atoi(e) : 128; }();      // A/B switch
    int c = 1;
    while (c < 64 && nblocks * c < target && npoints / (2 * c) >= piece) c *= 2;
    return c;
}

int vo_launch_ransac(vo_ctx* ctx, const double* d_xy, vo_strides sxy, const double* d_X, vo_strides sX, int M,
                     const double* d_K, const vo_ransac_opts* o, const int32_t* d_sample_idx,
                     const double* d_poses_in, vo_ransac_work* w, void* d_out, uint8_t* d_inliers, const int32_t* d_M, int score_all,
                     double* d_pose_out) {
    const int H = o->num_hyp;
    if (H < 1 || M < 4 || !(o->max_reproj_err > 0)) { ctx->last_error = "ransac: need num_hyp >= 1, M >= 4, max_reproj_err > 0"; return VO_ERR_BAD_ARG; }
    if (o->mode != 0 && o->mode != 1) { ctx->last_error = "ransac: mode must be 0 or 1"; return VO_ERR_BAD_ARG; }
    const double thr2 = o->max_reproj_err * o->max_reproj_err;
    const double* Rt = d_poses_in ? d_poses_in : w->Rt;
    // Two stages when the run is long.  The first generates, fully scores and scans the first RANSAC_STAGE hypotheses.
    // With the adaptive stop (confidence > 0) the sequential semantics usually end among them -- the record at the
    // stop depends on the hypotheses before it only -- and the second stage returns at once.  Otherwise the second
    // stage scores the rest on the points up to a first mark, leaves out the poses that can no longer beat the first
    // stage's record (k_survivors), scores the survivors up to a second mark, sieves again, scores what is left on
    // the remaining points, and scans all H.  The marks are chosen on the device from the first stage's record
    // (k_best).  Same record, trial count and inlier set as one stage over everything.
    // score_all: the caller reads every hypothesis' counts -- one stage, nothing left out.
    const int32_t* stopped = (const int32_t*)((const double*)d_out + 12) + 5;
    const bool staged = H > 2 * RANSAC_STAGE && !score_all && w->bound && w->list && M >= 4096;
    // without a confidence nothing can end the run early: a short first stage is enough to set the bound, and all
    // hypotheses are generated in one launch
    const int S = o->confidence > 0 ? RANSAC_STAGE : RANSAC_STAGE / 4;
    // counts, costs, bound, list lengths start at zero: cleared by the hypothesis generator (every hypothesis its own
    // sums), or, when the caller supplies the poses and no generator runs, by a launch of its own
    const bool zero_in_hyp = !d_poses_in && w->bound;
    if (!zero_in_hyp) {
        const long nwords = (long)(w->zero_bytes / 4);
        hipLaunchKernelGGL(k_zero, dim3((unsigned)((nwords + 63) / 64)), dim3(64), 0, ctx->stream, (unsigned*)w->cnt, nwords);
    }
    auto hyp = [&](int h0, int h1, bool later) {
        if (d_poses_in) return;
        HypArgs h;
        h.xy = d_xy; h.sxy = sxy; h.X = d_X; h.sX = sX; h.M = M; h.K = d_K; h.H = h1; h.seed = o->seed;
        h.sample_idx = d_sample_idx; h.Rt = w->Rt; h.h0 = h0; h.stopped = later ? stopped : nullptr; h.M_dev = d_M;
        h.zcnt = zero_in_hyp ? w->cnt : nullptr; h.zcost = w->cost; h.zblock = w->bound;
        hipLaunchKernelGGL(k_hyp, dim3((h1 - h0 + 63) / 64), dim3(64), 0, ctx->stream, h);
    };
    // the sieve marks live behind the bound and the survivor counts in the run's 256-byte block (zeroed with it)
    int32_t* const sieve = w->bound ? (int32_t*)(w->bound + 4) : nullptr;
    // marks: q0 / q1 index the sieve marks {0, first, second, all} instead of counting 64ths
    auto score = [&](int h0, int h1, int q0, int q1, bool later, int from_list, bool marks) {   // from_list: 0 no, 1 / 2 which list
        ScoreArgs s;
        s.xy = d_xy; s.sxy = sxy; s.X = d_X; s.sX = sX; s.M = M; s.K = d_K; s.H = h1; s.thr2 = thr2; s.Rt = Rt;
        s.cnt = w->cnt; s.cost = w->cost; s.counters = ctx->d_counters; s.h0 = h0; s.stopped = later ? stopped : nullptr; s.M_dev = d_M;
        s.list = from_list == 0 ? nullptr : (from_list == 1 ? w->list : w->list + 4 * (size_t)H);
        s.list_n = w->list_n + (from_list == 2 ? 1 : 0); s.q0 = q0; s.q1 = q1; s.qdev = marks ? sieve : nullptr;
        s.tests = ctx->time_score ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(vo_int_head_words(ctx)) + 64) : nullptr;
        const int nb = (h1 - h0 + SC_HB - 1) / SC_HB;
        // survivor launches: few blocks do anything (how few is known on the device only), so each takes a short
        // piece of the range -- the launch is as long as one block
        // points of the range: known here only without marks; with them about a third of the points
        const int npts = marks ? M / 3 : (int)((long)M * (q1 - q0) / 64);
        const int chunks = from_list ? (npts >= 16 * 512 ? 16 : (npts >= 2048 ? 4 : 1)) : score_chunks(nb, npts);
        // vo_ctx_time_score: a lap of the ctx's event stopwatch around every k_score launch (bench.py's roofline entry)
        if (ctx->time_score) (void)vo_timer_lap_begin(ctx);
        hipLaunchKernelGGL(k_score, dim3(nb, chunks), dim3(SC_NT), 0, ctx->stream, s);
        if (ctx->time_score) (void)vo_timer_lap_end(ctx);
    };
    auto best = [&](int h1, int later, bool write_bound) {
        BestArgs b;
        b.cnt = w->cnt; b.cost = w->cost; b.Rt = Rt; b.H = h1; b.M = M; b.mode = o->mode; b.min_inl = o->min_inliers;
        b.confidence = o->confidence; b.out = (double*)d_out; b.sample_size = 3; b.later_stage = later; b.M_dev = d_M;
        b.bound_out = write_bound ? w->bound : nullptr; b.sieve_out = sieve;
        b.qthr = (unsigned long long)(thr2 * 1048576.0);
        if (ctx->in_lane) hipLaunchKernelGGL(k_best<64>, dim3(1), dim3(64), 0, ctx->stream, b);
        else hipLaunchKernelGGL(k_best<256>, dim3(1), dim3(256), 0, ctx->stream, b);
    };
    // few points, long run, adaptive stop (the frame loop: 300 landmarks, 2000 trials, 99 %): no sieve -- there is too
    // little to leave out -- but the prefix first all the same: the run usually ends inside it, and the launches over
    // the other hypotheses return at once instead of generating and scoring them
    const bool prefix = !staged && H > 2 * RANSAC_STAGE && !score_all && o->confidence > 0 && !d_poses_in;
    // ... and when the points are few, a stage is one launch (k_pose_small)
    auto small_stage = [&](int h0, int h1, bool later) {
        SmallArgs sa;
        HypArgs& h = sa.h;
        h.xy = d_xy; h.sxy = sxy; h.X = d_X; h.sX = sX; h.M = M; h.K = d_K; h.H = h1; h.seed = o->seed;
        h.sample_idx = d_sample_idx; h.Rt = w->Rt; h.h0 = h0; h.stopped = later ? stopped : nullptr; h.M_dev = d_M;
        h.zcnt = nullptr; h.zcost = nullptr; h.zblock = nullptr;
        ScoreArgs& sc = sa.s;
        sc.xy = d_xy; sc.sxy = sxy; sc.X = d_X; sc.sX = sX; sc.M = M; sc.K = d_K; sc.H = h1; sc.thr2 = thr2; sc.Rt = w->Rt;
        sc.cnt = w->cnt; sc.cost = w->cost; sc.counters = ctx->d_counters; sc.h0 = h0; sc.stopped = nullptr; sc.M_dev = d_M;
        sc.list = nullptr; sc.list_n = nullptr; sc.q0 = 0; sc.q1 = 64; sc.qdev = nullptr; sc.tests = nullptr;
        BestArgs& b = sa.b;
        b.cnt = w->cnt; b.cost = w->cost; b.Rt = w->Rt; b.H = h1; b.M = M; b.mode = o->mode; b.min_inl = o->min_inliers;
        b.confidence = o->confidence; b.out = (double*)d_out; b.sample_size = 3; b.later_stage = later ? 1 : 0; b.M_dev = d_M;
        b.bound_out = nullptr; b.sieve_out = nullptr; b.qthr = (unsigned long long)(thr2 * 1048576.0);
        sa.ticket = w->ticket; sa.nblocks = (h1 - h0 + SC_HB - 1) / SC_HB;
        hipLaunchKernelGGL(k_pose_small, dim3(sa.nblocks), dim3(64), 0, ctx->stream, sa);
    };
    if (prefix && M <= RANSAC_SMALL_M && w->ticket) {
        small_stage(0, S, false);
        small_stage(S, H, true);
    } else if (prefix) {
        hyp(0, S, false);
        score(0, S, 0, 64, false, 0, false);
        best(S, 0, false);
        hyp(S, H, true);
        score(S, H, 0, 64, true, 0, false);
        best(H, 1, false);
    } else if (!staged) {
        hyp(0, H, false);
        score(0, H, 0, 64, false, 0, false);
        best(H, 0, false);
    } else {
        hyp(0, o->confidence > 0 ? S : H, false);
        score(0, S, 0, 64, false, 0, false);
        best(S, 0, true);
        if (o->confidence > 0) hyp(S, H, true);
        score(S, H, 0, 1, true, 0, true);                     // the points up to the first mark
        SurvArgs sv;
        sv.cnt = w->cnt; sv.cost = w->cost; sv.bound = w->bound; sv.h0 = S; sv.H = H; sv.M = M; sv.mode = o->mode;
        sv.stopped = stopped; sv.M_dev = d_M; sv.qdev = sieve;
        sv.seen = 1; sv.in = nullptr; sv.in_n = nullptr; sv.list = w->list; sv.list_n = w->list_n;
        hipLaunchKernelGGL(k_survivors, dim3((4 * (H - S) + 63) / 64), dim3(64), 0, ctx->stream, sv);
        score(S, H, 1, 2, true, 1, true);                     // the survivors up to the second mark
        sv.seen = 2; sv.in = w->list; sv.in_n = w->list_n; sv.list = w->list + 4 * (size_t)H; sv.list_n = w->list_n + 1;
        hipLaunchKernelGGL(k_survivors, dim3((4 * (H - S) + 63) / 64), dim3(64), 0, ctx->stream, sv);
        score(S, H, 2, 3, true, 2, true);                     // what is left on the rest
        best(H, 1, false);
    }
    if (!d_inliers) {      // the caller's next launch does what k_inliers does (the frame loop: k_pose_tail)
        VO_HIP_CHECK(ctx, hipGetLastError());
        return VO_OK;
    }
    InlArgs in;
    in.xy = d_xy; in.sxy = sxy; in.X = d_X; in.sX = sX; in.M = M; in.K = d_K; in.thr2 = thr2; in.out = (double*)d_out;
    in.inliers = d_inliers; in.M_dev = d_M; in.pose_out = d_pose_out;
    hipLaunchKernelGGL(k_inliers, dim3((M + 63) / 64), dim3(64), 0, ctx->stream, in);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}


// ================= allocation-free variant of vo_dev_p3p_ransac =================
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// it): the per-hypothesis work arrays (poses, counts, MSAC costs) live in a file-scope device array instead of the
// ctx's grow-only buffer.  Every GPU the library runs on holds its own copy of the array at its own address, until
// the process ends; the address is looked up once per device (vo_static_scratch_resolve, called by vo_ctx_create
// with that device current) and from then on only the looked-up pointer is used -- the array's NAME is a host-side
// stand-in, not a device address.  All calls on one GPU share the array: calls that overlap on one GPU (two ctxs,
// two threads) race.  The launcher zeroes what must start at zero (counts, costs, the survivor count) with one
// small kernel (k_zero) on the looked-up pointer at the start of every run.
namespace {
constexpr size_t static_pad(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t STATIC_RT = static_pad((size_t)VO_STATIC_MAX_HYP * 48 * sizeof(double));
constexpr size_t STATIC_CNT = static_pad((size_t)VO_STATIC_MAX_HYP * 4 * sizeof(int32_t));
constexpr size_t STATIC_COST = static_pad((size_t)VO_STATIC_MAX_HYP * 4 * sizeof(unsigned long long));
constexpr size_t STATIC_LIST = static_pad((size_t)VO_STATIC_MAX_HYP * 8 * sizeof(int32_t));   // two survivor lists
constexpr size_t STATIC_BYTES = STATIC_RT + STATIC_CNT + STATIC_COST + 256 + STATIC_LIST;   // 2.4 MB for 5120 hypotheses
constexpr int STATIC_MAX_DEVICES = 64;
}
// (no __attribute__((used)) here: the host-side use below is what makes the compiler keep the array in the code
// object AND give it a global symbol the runtime can look up; with the attribute it stays a local symbol and
// hipGetSymbolAddress aborts)
static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];
// per device: where that GPU's copy of `scratch` lives.  Atomic: contexts are created from any thread, and every thread
// that resolves a device's entry stores the same address
static std::atomic<unsigned char*> g_scratch_ptr[STATIC_MAX_DEVICES];

int vo_static_scratch_resolve(int device) {
    if (device < 0 || device >= STATIC_MAX_DEVICES) return VO_ERR_BAD_ARG;
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return VO_OK;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return VO_ERR_HIP;   // the current device's copy
    g_scratch_ptr[device].store((unsigned char*)p, std::memory_order_release);
    return VO_OK;
}

int vo_dev_p3p_ransac_static(vo_ctx* ctx, const double* d_xy, const double* d_X, int M, const double* d_K,
                             const vo_ransac_opts* o, const int32_t* d_sample_idx, const double* d_hyp_poses_in,
                             void* d_out, uint8_t* d_inliers) {
    if (!ctx || !o || !d_xy || !d_X || !d_K || !d_out || !d_inliers) return VO_ERR_BAD_ARG;
    if (o->num_hyp < 1) { ctx->last_error = "ransac: num_hyp < 1"; return VO_ERR_BAD_ARG; }
    if (o->num_hyp > VO_STATIC_MAX_HYP) {
        ctx->last_error = "vo_dev_p3p_ransac_static: num_hyp exceeds VO_STATIC_MAX_HYP (the static work array holds 5120 "
                          "hypotheses); use vo_dev_p3p_ransac";
        return VO_ERR_UNSUPPORTED;
    }
    if (ctx->device < 0 || ctx->device >= STATIC_MAX_DEVICES) { ctx->last_error = "vo_dev_p3p_ransac_static: device index beyond the table"; return VO_ERR_UNSUPPORTED; }
    (void)hipSetDevice(ctx->device);
    if (!g_scratch_ptr[ctx->device].load(std::memory_order_acquire)) {   // vo_ctx_create resolves it; a ctx made before a reload of the table ends here
        const int rc = vo_static_scratch_resolve(ctx->device);
        if (rc) { ctx->last_error = "vo_dev_p3p_ransac_static: hipGetSymbolAddress failed"; return rc; }
    }
    unsigned char* base = g_scratch_ptr[ctx->device].load(std::memory_order_acquire);   // never the array's name
    vo_ransac_work w;
    w.Rt = (double*)base;
    w.cnt = (int32_t*)(base + STATIC_RT);
    w.cost = (unsigned long long*)(base + STATIC_RT + STATIC_CNT);
    w.bound = (unsigned long long*)(base + STATIC_RT + STATIC_CNT + STATIC_COST);
    w.list_n = (int32_t*)(w.bound + 2);
    w.zero_bytes = STATIC_CNT + STATIC_COST + 256;
    w.list = (int32_t*)(base + STATIC_RT + STATIC_CNT + STATIC_COST + 256);
    vo_strides s2 = {2, 1}, s3 = {3, 1};
    return vo_launch_ransac(ctx, d_xy, s2, d_X, s3, M, d_K, o, d_sample_idx, d_hyp_poses_in, &w, d_out, d_inliers);
}


// ================= essential-matrix MSAC (bootstrap, modules/bootstrap.m:76-79; oracle N10) =================
namespace {

// 8 distinct indices: index j is the (r_j mod (m - j))-th point not drawn before (essential.c)
__device__ void philox_sample8(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
    uint32_t r[8];
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int b = 0; b < 2; ++b) {
        uint32_t c[4] = {hyp, (uint32_t)b, 0, 0};
        philox4x32_10(c, key);
        for (int k = 0; k < 4; ++k) r[4 * b + k] = c[k];
    }
    uint32_t sorted[8];
    for (int j = 0; j < 8; ++j) {
        uint32_t v = r[j] % (m - (uint32_t)j);
        for (int k = 0; k < j; ++k)
            if (v >= sorted[k]) ++v;
        int pos = j;
        while (pos > 0 && sorted[pos - 1] > v) { sorted[pos] = sorted[pos - 1]; --pos; }
        sorted[pos] = v;
        idx[j] = (int32_t)v;
    }
}

// one-sided Jacobi on the columns of the m x n row-major A (A V = U S), fixed sweeps and pair order
__device__ void hestenes(double* A, int m, int n, double* V, int sweeps) {
    for (int i = 0; i < n * n; ++i) V[i] = 0.0;
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
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < m; ++k) {
                    const double ap = A[k * n + p], aq = A[k * n + q];
                    A[k * n + p] = c * ap - sn * aq; A[k * n + q] = sn * ap + c * aq;
                }
                for (int k = 0; k < n; ++k) {
                    const double vp = V[k * n + p], vq = V[k * n + q];
                    V[k * n + p] = c * vp - sn * vq; V[k * n + q] = sn * vp + c * vq;
                }
            }
}
__device__ double colnorm2(const double* A, int m, int n, int j) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += A[k * n + j] * A[k * n + j];
    return s;
}
__device__ void hartley8(const double* x, const double* y, double* xn, double* yn, double* T) {
    double cx = 0, cy = 0;
    for (int i = 0; i < 8; ++i) { cx += x[i]; cy += y[i]; }
    cx /= 8.0; cy /= 8.0;
    double d = 0;
    for (int i = 0; i < 8; ++i) { const double dx = x[i] - cx, dy = y[i] - cy; d += sqrt(dx * dx + dy * dy); }
    d /= 8.0;
    const double s = 1.4142135623730951 / d;
    for (int i = 0; i < 8; ++i) { xn[i] = (x[i] - cx) * s; yn[i] = (y[i] - cy) * s; }
    T[0] = s; T[1] = 0; T[2] = -(s * cx); T[3] = 0; T[4] = s; T[5] = -(s * cy); T[6] = 0; T[7] = 0; T[8] = 1;
}
__device__ void mat3(const double* A, const double* B, double* C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}
__device__ void mat3_tn(const double* A, const double* B, double* C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r] * B[c] + A[3 + r] * B[3 + c]) + A[6 + r] * B[6 + c];
}
__device__ int essential_project(const double* E0, double* E) {
    double A[9], V[9];
    for (int k = 0; k < 9; ++k) A[k] = E0[k];
    hestenes(A, 3, 3, V, 16);
    const double nn[3] = {colnorm2(A, 3, 3, 0), colnorm2(A, 3, 3, 1), colnorm2(A, 3, 3, 2)};
    int a = 0;
    if (nn[1] > nn[a]) a = 1;
    if (nn[2] > nn[a]) a = 2;
    int b = a == 0 ? 1 : 0;
    for (int j = 0; j < 3; ++j) if (j != a && nn[j] > nn[b]) b = j;
    if (!(nn[b] > 0.0)) { for (int k = 0; k < 9; ++k) E[k] = 0.0; return 0; }
    const double sa = sqrt(nn[a]), sb = sqrt(nn[b]);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k)
            E[r * 3 + k] = (A[r * 3 + a] / sa) * V[k * 3 + a] + (A[r * 3 + b] / sb) * V[k * 3 + b];
    return 1;
}
__device__ int eight_point(const double* x0, const double* y0, const double* x1, const double* y1, double* E) {
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
    for (int j = 0; j < 9; ++j) { const double n2 = colnorm2(A, 8, 9, j); if (n2 < bn) { bn = n2; best = j; } }
    double F[9], FT0[9], E0[9];
    for (int k = 0; k < 9; ++k) F[k] = V[k * 9 + best];
    mat3(F, T0, FT0);
    mat3_tn(T1, FT0, E0);
    bool finite = true;
    for (int k = 0; k < 9; ++k) if (!(fabs(E0[k]) < INFINITY)) finite = false;
    if (!finite) { for (int k = 0; k < 9; ++k) E[k] = 0.0; return 0; }
    return essential_project(E0, E);
}
__device__ void fundamental_from_essential(const double* E, const double* K, double* F) {
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const double Ki[9] = {1.0 / fx, -sk / (fx * fy), (sk * cy - cx * fy) / (fx * fy), 0, 1.0 / fy, -cy / fy, 0, 0, 1};
    double EK[9];
    mat3(E, Ki, EK);
    mat3_tn(Ki, EK, F);
}
__device__ __forceinline__ double sampson(const double* F, double u0, double v0, double u1, double v1) {
    const double a0 = (F[0] * u0 + F[1] * v0) + F[2], a1 = (F[3] * u0 + F[4] * v0) + F[5], a2 = (F[6] * u0 + F[7] * v0) + F[8];
    const double b0 = (F[0] * u1 + F[3] * v1) + F[6], b1 = (F[1] * u1 + F[4] * v1) + F[7];
    const double num = (u1 * a0 + v1 * a1) + a2;
    const double den = (a0 * a0 + a1 * a1) + (b0 * b0 + b1 * b1);
    return (num * num) / den;
}

struct EHypArgs {
    const double *p0, *p1; vo_strides sp; int M; const double* K; int H; uint64_t seed; const int32_t* sample_idx;
    double* E48;   // hypothesis h at E48 + 48 h (9 doubles; the stride matches the pose table k_best reads)
};
__global__ __launch_bounds__(64) void k_e_hyp(EHypArgs a) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= a.H) return;
    int32_t id[8];
    bool ok = true;
    if (a.sample_idx) {
        for (int k = 0; k < 8; ++k) { id[k] = a.sample_idx[8 * h + k]; if (id[k] < 0 || id[k] >= a.M) ok = false; }
    } else philox_sample8(a.seed, (uint32_t)h, (uint32_t)a.M, id);
    double E[9];
    for (int k = 0; k < 9; ++k) E[k] = 0.0;
    if (ok) {
        const double fx = a.K[0], sk = a.K[1], cx = a.K[2], fy = a.K[4], cy = a.K[5];
        double x0[8], y0[8], x1[8], y1[8];
        for (int k = 0; k < 8; ++k) {
            const long i = id[k];
            const double u0 = a.p0[i * a.sp.xi], v0 = a.p0[i * a.sp.xi + a.sp.xc];
            const double u1 = a.p1[i * a.sp.xi], v1 = a.p1[i * a.sp.xi + a.sp.xc];
            y0[k] = (v0 - cy) / fy; x0[k] = ((u0 - cx) - sk * y0[k]) / fx;
            y1[k] = (v1 - cy) / fy; x1[k] = ((u1 - cx) - sk * y1[k]) / fx;
        }
        eight_point(x0, y0, x1, y1, E);
    }
    double* o = a.E48 + 48 * (long)h;
    for (int k = 0; k < 9; ++k) o[k] = E[k];
    for (int k = 9; k < 48; ++k) o[k] = 0.0;
}

// ---- Nister's five-point solver (definition N10b, oracle/five_point.c: the same operations in the same order) ----
__device__ void philox_sample5(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx) {
    uint32_t r[8], c[4], key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    c[0] = hyp; c[1] = 0; c[2] = 0; c[3] = 0; philox4x32_10(c, key); for (int k = 0; k < 4; ++k) r[k] = c[k];
    c[0] = hyp; c[1] = 1; c[2] = 0; c[3] = 0; philox4x32_10(c, key); for (int k = 0; k < 4; ++k) r[4 + k] = c[k];
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
// monomials.  degree <= 1: [x y z 1]; degree <= 2: [x^2 xy xz y^2 yz z^2 x y z 1]; degree <= 3: Nister's order
// [x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1]
__device__ const int FP_M11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
__device__ const int FP_M21[10][4] = {{0, 2, 4, 5}, {2, 3, 8, 9}, {4, 8, 10, 11}, {3, 1, 6, 7}, {8, 6, 13, 14},
                                      {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};
__device__ void fp_acc11(double* o, const double* a, const double* b, double s) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[FP_M11[i][j]] += s * (a[i] * b[j]);
}
__device__ void fp_acc21(double* o, const double* a, const double* b, double s) {
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) o[FP_M21[i][j]] += s * (a[i] * b[j]);
}
__device__ double fp_horner(const double* c, int n, double z) {
    double v = c[n];
    for (int k = n - 1; k >= 0; --k) v = v * z + c[k];
    return v;
}
__device__ void fp_zmul(const double* a, int na, const double* b, int nb, double* o) {
    for (int k = 0; k <= na + nb; ++k) o[k] = 0.0;
    for (int i = 0; i <= na; ++i)
        for (int j = 0; j <= nb; ++j) o[i + j] += a[i] * b[j];
}
// the simple real roots of c[0] + ... + c[n] z^n inside the Cauchy bound, ascending: bisection between the roots of the
// derivative, from the linear one upwards
__device__ int fp_real_roots(const double* c, int n, double* roots) {
    while (n > 0 && c[n] == 0.0) --n;
    if (n < 1) return 0;
    double bound = 0.0;
    for (int k = 0; k < n; ++k) { const double r = fabs(c[k] / c[n]); if (r > bound) bound = r; }
    bound += 1.0;
    if (!(bound < 1e12)) bound = 1e12;
    double d[11][11];
    for (int i = 0; i <= n; ++i) d[0][i] = c[i];
    for (int k = 1; k < n; ++k)
        for (int i = 0; i <= n - k; ++i) d[k][i] = d[k - 1][i + 1] * (double)(i + 1);
    double prev[10], cur[10];
    int np = 0;
    for (int k = n - 1; k >= 0; --k) {
        const int deg = n - k;
        int nc = 0;
        for (int s = 0; s <= np; ++s) {
            double lo = s == 0 ? -bound : prev[s - 1], hi = s == np ? bound : prev[s];
            const double flo = fp_horner(d[k], deg, lo), fhi = fp_horner(d[k], deg, hi);
            if (flo == 0.0) { if (s == 0) cur[nc++] = lo; continue; }
            if (fhi == 0.0) { cur[nc++] = hi; continue; }
            if ((flo < 0.0) == (fhi < 0.0)) continue;
            for (int it = 0; it < 200; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                const double fm = fp_horner(d[k], deg, mid);
                if (fm == 0.0) { lo = hi = mid; break; }
                if ((fm < 0.0) == (flo < 0.0)) lo = mid; else hi = mid;
            }
            cur[nc++] = 0.5 * (lo + hi);
        }
        np = nc;
        for (int s = 0; s < nc; ++s) prev[s] = cur[s];
    }
    for (int s = 0; s < np; ++s) roots[s] = prev[s];
    return np;
}
// 5 correspondences (K^-1-normalised) -> up to 10 matrices E, ascending in the root; returns their number
__device__ int five_point(const double* x0, const double* y0, const double* x1, const double* y1, double* Es) {
    double A[45], V[81];
    for (int i = 0; i < 5; ++i) {
        double* r = A + 9 * i;
        r[0] = x1[i] * x0[i]; r[1] = x1[i] * y0[i]; r[2] = x1[i];
        r[3] = y1[i] * x0[i]; r[4] = y1[i] * y0[i]; r[5] = y1[i];
        r[6] = x0[i]; r[7] = y0[i]; r[8] = 1.0;
    }
    hestenes(A, 5, 9, V, 16);
    double nn[9]; int null_col[4]; bool taken[9];
    for (int j = 0; j < 9; ++j) { taken[j] = false; nn[j] = colnorm2(A, 5, 9, j); }
    for (int t = 0; t < 4; ++t) {
        int best = -1;
        for (int j = 0; j < 9; ++j) if (!taken[j] && (best < 0 || nn[j] < nn[best])) best = j;
        taken[best] = true;
    }
    for (int j = 0, t = 0; j < 9; ++j) if (taken[j]) null_col[t++] = j;
    double E1[9][4];
    for (int e = 0; e < 9; ++e)
        for (int b = 0; b < 4; ++b) E1[e][b] = V[e * 9 + null_col[b]];
    double EEt[9][10], tr[10], cons[10][20];
    for (int e = 0; e < 9; ++e) for (int i = 0; i < 10; ++i) EEt[e][i] = 0.0;
    for (int e = 0; e < 10; ++e) for (int i = 0; i < 20; ++i) cons[e][i] = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) fp_acc11(EEt[3 * r + c], E1[3 * r + k], E1[3 * c + k], 1.0);
    for (int i = 0; i < 10; ++i) tr[i] = (EEt[0][i] + EEt[4][i]) + EEt[8][i];
    {
        double m0[10], m1[10], m2[10];
        for (int i = 0; i < 10; ++i) { m0[i] = 0.0; m1[i] = 0.0; m2[i] = 0.0; }
        fp_acc11(m0, E1[4], E1[8], 1.0); fp_acc11(m0, E1[5], E1[7], -1.0);
        fp_acc11(m1, E1[3], E1[8], 1.0); fp_acc11(m1, E1[5], E1[6], -1.0);
        fp_acc11(m2, E1[3], E1[7], 1.0); fp_acc11(m2, E1[4], E1[6], -1.0);
        fp_acc21(cons[0], m0, E1[0], 1.0); fp_acc21(cons[0], m1, E1[1], -1.0); fp_acc21(cons[0], m2, E1[2], 1.0);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double* o = cons[1 + 3 * r + c];
            for (int k = 0; k < 3; ++k) fp_acc21(o, EEt[3 * r + k], E1[3 * k + c], 2.0);
            fp_acc21(o, tr, E1[3 * r + c], -1.0);
        }
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
    double B[3][3][5];
    for (int q = 0; q < 3; ++q) {
        const double* e = cons[4 + 2 * q] + 10; const double* f = cons[5 + 2 * q] + 10;
        for (int part = 0; part < 2; ++part) {
            const double* pe = e + 3 * part; const double* pf = f + 3 * part;
            B[q][part][0] = pe[2]; B[q][part][1] = pe[1] - pf[2]; B[q][part][2] = pe[0] - pf[1]; B[q][part][3] = -pf[0]; B[q][part][4] = 0.0;
        }
        B[q][2][0] = e[9]; B[q][2][1] = e[8] - f[9]; B[q][2][2] = e[7] - f[8]; B[q][2][3] = e[6] - f[7]; B[q][2][4] = -f[6];
    }
    double t1[8], t2[8], c0[8], c1[8], c2[7], d0[11], d1[11], d2[11], det[11];
    fp_zmul(B[1][1], 3, B[2][2], 4, t1); fp_zmul(B[1][2], 4, B[2][1], 3, t2);
    for (int k = 0; k < 8; ++k) c0[k] = t1[k] - t2[k];
    fp_zmul(B[1][0], 3, B[2][2], 4, t1); fp_zmul(B[1][2], 4, B[2][0], 3, t2);
    for (int k = 0; k < 8; ++k) c1[k] = t1[k] - t2[k];
    fp_zmul(B[1][0], 3, B[2][1], 3, t1); fp_zmul(B[1][1], 3, B[2][0], 3, t2);
    for (int k = 0; k < 7; ++k) c2[k] = t1[k] - t2[k];
    fp_zmul(B[0][0], 3, c0, 7, d0); fp_zmul(B[0][1], 3, c1, 7, d1); fp_zmul(B[0][2], 4, c2, 6, d2);
    for (int k = 0; k < 11; ++k) det[k] = (d0[k] - d1[k]) + d2[k];
    for (int k = 0; k < 11; ++k) if (!(fabs(det[k]) < INFINITY)) return 0;
    double roots[10];
    const int nr = fp_real_roots(det, 10, roots);
    int n = 0;
    for (int s = 0; s < nr; ++s) {
        const double z = roots[s];
        double bx[3], by[3], b1[3];
        for (int q = 0; q < 3; ++q) { bx[q] = fp_horner(B[q][0], 3, z); by[q] = fp_horner(B[q][1], 3, z); b1[q] = fp_horner(B[q][2], 4, z); }
        int pa = 0, pb = 1;
        double den = bx[0] * by[1] - by[0] * bx[1];
        {
            const double d02 = bx[0] * by[2] - by[0] * bx[2], d12 = bx[1] * by[2] - by[1] * bx[2];
            if (fabs(d02) > fabs(den)) { den = d02; pa = 0; pb = 2; }
            if (fabs(d12) > fabs(den)) { den = d12; pa = 1; pb = 2; }
        }
        const double x = (by[pa] * b1[pb] - b1[pa] * by[pb]) / den, y = (b1[pa] * bx[pb] - bx[pa] * b1[pb]) / den;
        double E0[9];
        bool finite = true;
        for (int e = 0; e < 9; ++e) {
            E0[e] = ((x * E1[e][0] + y * E1[e][1]) + z * E1[e][2]) + E1[e][3];
            if (!(fabs(E0[e]) < INFINITY)) finite = false;
        }
        if (!finite) continue;
        if (essential_project(E0, Es + 9 * n)) ++n;
    }
    return n;
}
// one thread per sample; its (up to) ten solutions are rows 10 h .. 10 h + 9 of the table (unused rows: E = 0)
__global__ __launch_bounds__(64) void k_e5_hyp(EHypArgs a) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= a.H) return;
    int32_t id[5];
    bool ok = true;
    if (a.sample_idx) {
        for (int k = 0; k < 5; ++k) { id[k] = a.sample_idx[5 * h + k]; if (id[k] < 0 || id[k] >= a.M) ok = false; }
    } else philox_sample5(a.seed, (uint32_t)h, (uint32_t)a.M, id);
    double Es[90];
    int n = 0;
    if (ok) {
        const double fx = a.K[0], sk = a.K[1], cx = a.K[2], fy = a.K[4], cy = a.K[5];
        double x0[5], y0[5], x1[5], y1[5];
        for (int k = 0; k < 5; ++k) {
            const long i = id[k];
            const double u0 = a.p0[i * a.sp.xi], v0 = a.p0[i * a.sp.xi + a.sp.xc];
            const double u1 = a.p1[i * a.sp.xi], v1 = a.p1[i * a.sp.xi + a.sp.xc];
            y0[k] = (v0 - cy) / fy; x0[k] = ((u0 - cx) - sk * y0[k]) / fx;
            y1[k] = (v1 - cy) / fy; x1[k] = ((u1 - cx) - sk * y1[k]) / fx;
        }
        n = five_point(x0, y0, x1, y1, Es);
    }
    for (int s = 0; s < 10; ++s) {
        double* o = a.E48 + 48 * (10 * (long)h + s);
        for (int k = 0; k < 9; ++k) o[k] = s < n ? Es[9 * s + k] : 0.0;
        for (int k = 9; k < 48; ++k) o[k] = 0.0;
    }
}

struct EScoreArgs {
    const double *p0, *p1; vo_strides sp; int M; const double* K; int H; double thr; const double* E; int e_stride;
    int32_t* cnt; unsigned long long* cost; // H x 4: solution 0 carries the hypothesis, 1..3 are disabled
};
// one 256-thread block per hypothesis
__global__ __launch_bounds__(256) void k_e_score(EScoreArgs a) {
    __shared__ double sF[9];
    __shared__ int s_ok;
    __shared__ int s_cnt[4];
    __shared__ unsigned long long s_cost[4];
    const int h = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        const double* E = a.E + (long)a.e_stride * h;
        double e[9], F[9];
        bool nz = false;
        for (int k = 0; k < 9; ++k) { e[k] = E[k]; if (e[k] != 0.0) nz = true; }
        fundamental_from_essential(e, a.K, F);
        for (int k = 0; k < 9; ++k) sF[k] = F[k];
        s_ok = nz ? 1 : 0;
    }
    __syncthreads();
    const unsigned long long qthr = (unsigned long long)(a.thr * 1048576.0);
    int c = 0; unsigned long long q = 0;
    if (s_ok) {
        double F[9];
        for (int k = 0; k < 9; ++k) F[k] = sF[k];
        for (int i = tid; i < a.M; i += 256) {
            const double d = sampson(F, a.p0[i * a.sp.xi], a.p0[i * a.sp.xi + a.sp.xc], a.p1[i * a.sp.xi], a.p1[i * a.sp.xi + a.sp.xc]);
            if (d < a.thr) { ++c; q += (unsigned long long)(d * 1048576.0); } else q += qthr;
        }
    } else {
        for (int i = tid; i < a.M; i += 256) q += qthr;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { c += __shfl_xor(c, d, 64); q += __shfl_xor(q, d, 64); }
    if ((tid & 63) == 0) { s_cnt[tid >> 6] = c; s_cost[tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        a.cnt[4 * (long)h] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        a.cost[4 * (long)h] = (s_cost[0] + s_cost[1]) + (s_cost[2] + s_cost[3]);
        for (int s = 1; s < 4; ++s) { a.cnt[4 * (long)h + s] = 0; a.cost[4 * (long)h + s] = ~0ull; }
    }
}

struct EInlArgs { const double *p0, *p1; vo_strides sp; int M; const double* K; double thr; double* out; uint8_t* inliers; };
__global__ __launch_bounds__(256) void k_e_inliers(EInlArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int32_t* info = (int32_t*)(a.out + 12);
    bool in = false;
    if (i < a.M && info[0] == 0) {
        double F[9];
        fundamental_from_essential(a.out, a.K, F);
        in = sampson(F, a.p0[i * a.sp.xi], a.p0[i * a.sp.xi + a.sp.xc], a.p1[i * a.sp.xi], a.p1[i * a.sp.xi + a.sp.xc]) < a.thr;
    }
    if (i < a.M) a.inliers[i] = in ? 1 : 0;
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&info[4], (int)__popcll(bal));
}

} // namespace

// d_out: E (9 doubles, 3 of padding) followed by vo_ransac_info as 5 int32; w as for the pose step.
// d_E_in (optional): H x 9 hypotheses to score instead of sampling ("identical hypotheses").
int vo_launch_essential(vo_ctx* ctx, const double* d_p0, const double* d_p1, vo_strides sp, int M, const double* d_K,
                        const vo_ransac_opts* o, const int32_t* d_sample_idx, const double* d_E_in, vo_ransac_work* w,
                        void* d_out, uint8_t* d_inliers, int solver) {
    // solver 8: one matrix per sample (eight-point); 5: Nister's five-point solver, ten rows of the table per sample
    // (w is prepared for 10 x num_hyp rows)
    if (solver != 8 && solver != 5) { ctx->last_error = "essential: solver must be 8 or 5"; return VO_ERR_BAD_ARG; }
    const int rpt = (solver == 5 && !d_E_in) ? 10 : 1;
    const int H = o->num_hyp * rpt;
    if (o->num_hyp < 1 || M < solver || !(o->max_reproj_err > 0)) { ctx->last_error = "essential: need num_hyp >= 1, M >= the sample size, MaxDistance > 0"; return VO_ERR_BAD_ARG; }
    const double* E = w->Rt; int e_stride = 48;
    if (d_E_in) { E = d_E_in; e_stride = 9; }
    else {
        EHypArgs h;
        h.p0 = d_p0; h.p1 = d_p1; h.sp = sp; h.M = M; h.K = d_K; h.H = o->num_hyp; h.seed = o->seed; h.sample_idx = d_sample_idx; h.E48 = w->Rt;
        if (solver == 5) hipLaunchKernelGGL(k_e5_hyp, dim3((o->num_hyp + 63) / 64), dim3(64), 0, ctx->stream, h);
        else hipLaunchKernelGGL(k_e_hyp, dim3((H + 63) / 64), dim3(64), 0, ctx->stream, h);
    }
    EScoreArgs s;
    s.p0 = d_p0; s.p1 = d_p1; s.sp = sp; s.M = M; s.K = d_K; s.H = H; s.thr = o->max_reproj_err; s.E = E; s.e_stride = e_stride;
    s.cnt = w->cnt; s.cost = w->cost;
    hipLaunchKernelGGL(k_e_score, dim3(H), dim3(256), 0, ctx->stream, s);
    if (d_E_in) { // k_best copies the winner out of a stride-48 table
        VO_HIP_CHECK(ctx, hipMemsetAsync(w->Rt, 0, sizeof(double) * 48 * (size_t)H, ctx->stream));
        VO_HIP_CHECK(ctx, hipMemcpy2DAsync(w->Rt, 48 * sizeof(double), d_E_in, 9 * sizeof(double), 9 * sizeof(double), H,
                                           hipMemcpyDeviceToDevice, ctx->stream));
    }
    BestArgs b;
    b.bound_out = nullptr; b.sieve_out = nullptr; b.qthr = 0;
    b.cnt = w->cnt; b.cost = w->cost; b.Rt = w->Rt; b.H = H; b.M = M; b.mode = 1; b.min_inl = o->min_inliers;
    b.confidence = o->confidence; b.out = (double*)d_out; b.sample_size = solver; b.later_stage = 0; b.M_dev = nullptr;
    b.rows_per_trial = rpt;
    hipLaunchKernelGGL(k_best<256>, dim3(1), dim3(256), 0, ctx->stream, b);
    EInlArgs in;
    in.p0 = d_p0; in.p1 = d_p1; in.sp = sp; in.M = M; in.K = d_K; in.thr = o->max_reproj_err; in.out = (double*)d_out; in.inliers = d_inliers;
    hipLaunchKernelGGL(k_e_inliers, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, in);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_p3p_batch(vo_ctx* ctx, const double* d_P, const double* d_f, int n, double* d_poses) {
    if (n <= 0) return VO_OK;
    hipLaunchKernelGGL(k_p3p_batch, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_P, d_f, n, d_poses);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
