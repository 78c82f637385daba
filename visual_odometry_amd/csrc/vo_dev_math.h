// vo_dev_math.h -- device functions shared by translation units that are compiled with -ffp-contract=off
// (vo_geom.hip, vo_ransac.hip, vo_frame.hip): the same source gives the same bits wherever it is inlined.
#pragma once
#include <hip/hip_runtime.h>

// ---- pose / camera conventions (oracle/geom.c header; modules/VisualOdometry.m:48-61) ----
static __device__ __forceinline__ void load43(const double* p, int col_major, double* out) {
    if (col_major) {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 3; ++c) out[r * 3 + c] = p[c * 4 + r];
    } else {
        for (int k = 0; k < 12; ++k) out[k] = p[k];
    }
}

static __device__ __forceinline__ void camera_matrix(const double* pose, const double* K, double* cam) {
    double E[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[i * 3 + j] = pose[j * 3 + i];
    for (int j = 0; j < 3; ++j)
        E[9 + j] = -((pose[9] * pose[j * 3 + 0] + pose[10] * pose[j * 3 + 1]) + pose[11] * pose[j * 3 + 2]);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            cam[i * 3 + j] = (E[i * 3 + 0] * K[j * 3 + 0] + E[i * 3 + 1] * K[j * 3 + 1]) + E[i * 3 + 2] * K[j * 3 + 2];
}

static __device__ __forceinline__ void project(const double* cam, const double* X, double& u, double& v, double& w) {
    double pu = ((X[0] * cam[0] + X[1] * cam[3]) + X[2] * cam[6]) + cam[9];
    double pv = ((X[0] * cam[1] + X[1] * cam[4]) + X[2] * cam[7]) + cam[10];
    double pw = ((X[0] * cam[2] + X[1] * cam[5]) + X[2] * cam[8]) + cam[11];
    u = pu / pw; v = pv / pw; w = pw;
}


// ---- the pose step's inlier test (oracle/vo_oracle.h N9, old/ransacLocalization.m:96-99) ----
// Scoring arithmetic (oracle/vo_oracle.h N9).  Each pose is folded once into the 3x4 projection
// Pm = K [R_cw | t_cw]; the projection is kept homogeneous, so the inlier test needs no division:
//   ez = (u*z - u~)^2 + (v*z - v~)^2 < thr^2 * z^2,
// with explicit fused multiply-adds in a fixed order.  Only inliers pay for e2 = ez / z^2 (MSAC term).
static __device__ __forceinline__ double proj_elem(const double* Rt, const double* K, int r, int c) {
    const double g0 = c < 3 ? Rt[c] : Rt[9], g1 = c < 3 ? Rt[3 + c] : Rt[10], g2 = c < 3 ? Rt[6 + c] : Rt[11];
    if (r == 0) return fma(K[2], g2, fma(K[1], g1, K[0] * g0));
    if (r == 1) return fma(K[5], g2, K[4] * g1);
    return g2;
}
struct ProjT { double ez, z2; };
static __device__ __forceinline__ ProjT proj_terms(const double* P, double X0, double X1, double X2, double u, double v) {
    const double up = fma(P[2], X2, fma(P[1], X1, fma(P[0], X0, P[3])));
    const double vp = fma(P[6], X2, fma(P[5], X1, fma(P[4], X0, P[7])));
    const double zc = fma(P[10], X2, fma(P[9], X1, fma(P[8], X0, P[11])));
    const double du = fma(u, zc, -up), dv = fma(v, zc, -vp);
    ProjT r;
    r.ez = fma(dv, dv, du * du);
    r.z2 = zc * zc;
    return r;
}

