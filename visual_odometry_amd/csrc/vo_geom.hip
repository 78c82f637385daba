// vo_geom.hip -- per-candidate geometry, one thread per candidate, IEEE double in the oracle's
// operation order (compiled with -ffp-contract=off): two-view linear DLT triangulation
// (modules/VisualOdometry.m:93-101, old/linearTriangulation.m:28-39), reprojection error
// (modules/VisualOdometry.m:48-61), bearing angle (utils/calculateAngleDeg.m:7-10) and cheirality
// (modules/isInFrontOfCamera.m:8-11).  Conventions: oracle/geom.c header.
//
// Bound: ~60 algorithmic bytes and ~1.5 kflop (FP64) per candidate: launch-latency / HBM bound.
#include "vo_common.h"
#include "vo_dev_math.h"

namespace {

struct M43 { double m[12]; }; // 4x3 row-major

__device__ __forceinline__ double angle_deg(const double* X, const double* c0, const double* c1) {
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = X[i] - c0[i]; b[i] = X[i] - c1[i]; }
    double dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    double na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    double nb = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
    double c = dot / (na * nb);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    return acos(c) * (180.0 / 3.14159265358979323846);
}

// One-sided Jacobi as oracle/geom.c svd4_last defines it: a sweep is three rounds of two disjoint column pairs,
// (0,1)(2,3) / (0,2)(1,3) / (0,3)(1,2); it ends with the first sweep all of whose rotations were the identity up
// to rounding (c == 1.0), after 12 sweeps at the latest (5-6 on triangulation matrices).  The right-singular vector
// of the smallest column norm is the result.
// On TWO lanes per matrix: a thread alone on its SIMD issues one instruction every ~8 cycles whatever the
// dependencies, so the kernel's time is a thread's instruction count, and the two rotations of a round are
// independent.  Lane 0 of a pair keeps column 0 and its partner of the round, lane 1 the other two; between rounds
// one column (of A and of V) changes hands through a DPP quad permute.  Every rotation is the oracle's expression on
// the values a one-thread form would hold: same bits (the one-thread form this replaced gave the same outputs).
struct JCol { double a[4], v[4]; };
__device__ __forceinline__ double pair_xchg(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the other lane of the pair
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ JCol pair_xchg(const JCol& c) {
    JCol o;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.a[k] = pair_xchg(c.a[k]); o.v[k] = pair_xchg(c.v[k]); }
    return o;
}
__device__ __forceinline__ bool jacobi_pair_rotate(JCol& P, JCol& Q) {
    double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        alpha += P.a[k] * P.a[k];
        beta += Q.a[k] * Q.a[k];
        gamma += P.a[k] * Q.a[k];
    }
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    double c = 1.0 / sqrt(1.0 + t * t);
    double sn = c * t;
    if (gamma == 0.0) { c = 1.0; sn = 0.0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double ap = P.a[k], aq = Q.a[k];
        P.a[k] = c * ap - sn * aq; Q.a[k] = sn * ap + c * aq;
        const double vp = P.v[k], vq = Q.v[k];
        P.v[k] = c * vp - sn * vq; Q.v[k] = sn * vp + c * vq;
    }
    return c != 1.0;
}
// A: the full 4x4 matrix (both lanes hold it), role: 0 / 1 = the lane of the pair; vout on both lanes
__device__ __forceinline__ void svd4_last_pair(const double* A, int role, double* vout) {
    JCol P, Q;   // lane 0: columns (0,1); lane 1: (2,3)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        P.a[k] = role ? A[k * 4 + 2] : A[k * 4 + 0];
        Q.a[k] = role ? A[k * 4 + 3] : A[k * 4 + 1];
        P.v[k] = (k == (role ? 2 : 0)) ? 1.0 : 0.0;
        Q.v[k] = (k == (role ? 3 : 1)) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rot = jacobi_pair_rotate(P, Q);                        // (0,1) (2,3)
        JCol got = pair_xchg(role ? P : Q);                         // lane 0 gives column 1, lane 1 column 2
        if (role) P = got; else Q = got;                            // (0,2) (1,3)
        rot |= jacobi_pair_rotate(P, Q);
        got = pair_xchg(Q);                                         // columns 2 and 3 change sides
        Q = got;                                                    // (0,3) (1,2)
        rot |= jacobi_pair_rotate(P, Q);
        got = pair_xchg(role ? P : Q);                              // lane 0 gives column 3, lane 1 column 1
        if (role) { P = Q; Q = got; } else Q = got;                 // (0,1) (2,3) again
        const int other = __builtin_amdgcn_mov_dpp((int)rot, 0xB1, 0xF, 0xF, true);
        if (!(rot || other)) break;
    }
    double nP = 0, nQ = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { nP += P.a[k] * P.a[k]; nQ += Q.a[k] * Q.a[k]; }
    const double oP = pair_xchg(nP), oQ = pair_xchg(nQ);
    const double nn[4] = {role ? oP : nP, role ? oQ : nQ, role ? nP : oP, role ? nQ : oQ};
    int best = 0; double bn = INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (nn[j] < bn) { bn = nn[j]; best = j; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mine = (best & 1) ? Q.v[k] : P.v[k];
        const double theirs = pair_xchg(mine);
        vout[k] = (best >> 1) == role ? mine : theirs;
    }
}

struct TriArgs {
    const double *xy0, *xy1; vo_strides sxy;
    const double* poses0; int pose_col_major; const int32_t* pose_idx; int C, P;
    const double *pose1, *K;
    double* X; vo_strides sX; double *reproj, *angle; uint8_t* valid;
    const int32_t* C_dev;   // optional: the candidate count on the device, C its upper bound
};

__global__ __launch_bounds__(256) void k_triangulate(TriArgs a) {
    __shared__ double s_cam1[12], s_pose1[12], s_K[9];
    if (threadIdx.x == 0) {
        double p1[12], K[9];
        load43(a.pose1, a.pose_col_major, p1);
        for (int k = 0; k < 9; ++k) K[k] = a.K[k];
        double cam[12];
        camera_matrix(p1, K, cam);
        for (int k = 0; k < 12; ++k) { s_cam1[k] = cam[k]; s_pose1[k] = p1[k]; }
        for (int k = 0; k < 9; ++k) s_K[k] = K[k];
    }
    __syncthreads();
    const int i = blockIdx.x * 128 + (threadIdx.x >> 1);   // two lanes per candidate (svd4_last_pair)
    const int role = threadIdx.x & 1;
    if (i >= (a.C_dev ? min(a.C, *a.C_dev) : a.C)) return;
    int pi = a.pose_idx ? a.pose_idx[i] : i;
    if (pi < 0 || pi >= a.P) { // invalid table index: flag the candidate instead of reading out of bounds
        if (role == 0) {
            a.valid[i] = 0; a.reproj[i] = __builtin_nan(""); a.angle[i] = __builtin_nan("");
            for (int c = 0; c < 3; ++c) a.X[i * a.sX.xi + c * a.sX.xc] = __builtin_nan("");
        }
        return;
    }
    double pose0[12], K[9], cam0[12], cam1[12];
    load43(a.poses0 + 12 * (long)pi, a.pose_col_major, pose0);
    for (int k = 0; k < 9; ++k) K[k] = s_K[k];
    for (int k = 0; k < 12; ++k) cam1[k] = s_cam1[k];
    camera_matrix(pose0, K, cam0);
    const double x0 = a.xy0[i * a.sxy.xi], y0 = a.xy0[i * a.sxy.xi + a.sxy.xc];
    const double x1 = a.xy1[i * a.sxy.xi], y1 = a.xy1[i * a.sxy.xi + a.sxy.xc];
    double A[16], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0 * 4 + k] = x0 * cam0[k * 3 + 2] - cam0[k * 3 + 0];
        A[1 * 4 + k] = y0 * cam0[k * 3 + 2] - cam0[k * 3 + 1];
        A[2 * 4 + k] = x1 * cam1[k * 3 + 2] - cam1[k * 3 + 0];
        A[3 * 4 + k] = y1 * cam1[k * 3 + 2] - cam1[k * 3 + 1];
    }
    svd4_last_pair(A, role, h);
    if (role) return;
    double Xi[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
    double u, v, w0, w1;
    project(cam0, Xi, u, v, w0);
    const double e0 = sqrt((u - x0) * (u - x0) + (v - y0) * (v - y0));
    project(cam1, Xi, u, v, w1);
    const double e1 = sqrt((u - x1) * (u - x1) + (v - y1) * (v - y1));
    for (int c = 0; c < 3; ++c) a.X[i * a.sX.xi + c * a.sX.xc] = Xi[c];
    a.reproj[i] = 0.5 * (e0 + e1);
    a.valid[i] = (uint8_t)(w0 > 0.0 && w1 > 0.0);
    a.angle[i] = angle_deg(Xi, pose0 + 9, s_pose1 + 9);
}

// The toolbox call itself, `triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2)` as the reference makes
// it (modules/VisualOdometry.m:97-101, modules/candidateTriangulation.m:47-51, processFrame.m:55-59): n point pairs, the
// two 4 x 3 camera matrices given -- the same DLT rows, Jacobi and projections as k_triangulate, without the pose ->
// camera-matrix step and without the bearing angle (two lanes per pair, svd4_last_pair).
struct TriCamArgs {
    const double *xy0, *xy1; vo_strides sxy; const double *cam0, *cam1; int cam_col_major, n;
    double* X; vo_strides sX; double* reproj; uint8_t* valid;
};
__global__ __launch_bounds__(256) void k_triangulate_cam(TriCamArgs a) {
    __shared__ double s_cam[24];
    if (threadIdx.x < 2) {
        double c[12];
        load43(threadIdx.x ? a.cam1 : a.cam0, a.cam_col_major, c);
        for (int k = 0; k < 12; ++k) s_cam[12 * threadIdx.x + k] = c[k];
    }
    __syncthreads();
    const int i = blockIdx.x * 128 + (threadIdx.x >> 1);
    const int role = threadIdx.x & 1;
    if (i >= a.n) return;
    double cam0[12], cam1[12];
    for (int k = 0; k < 12; ++k) { cam0[k] = s_cam[k]; cam1[k] = s_cam[12 + k]; }
    const double x0 = a.xy0[i * a.sxy.xi], y0 = a.xy0[i * a.sxy.xi + a.sxy.xc];
    const double x1 = a.xy1[i * a.sxy.xi], y1 = a.xy1[i * a.sxy.xi + a.sxy.xc];
    double A[16], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0 * 4 + k] = x0 * cam0[k * 3 + 2] - cam0[k * 3 + 0];
        A[1 * 4 + k] = y0 * cam0[k * 3 + 2] - cam0[k * 3 + 1];
        A[2 * 4 + k] = x1 * cam1[k * 3 + 2] - cam1[k * 3 + 0];
        A[3 * 4 + k] = y1 * cam1[k * 3 + 2] - cam1[k * 3 + 1];
    }
    svd4_last_pair(A, role, h);
    if (role) return;
    double Xi[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
    double u, v, w0, w1;
    project(cam0, Xi, u, v, w0);
    const double e0 = sqrt((u - x0) * (u - x0) + (v - y0) * (v - y0));
    project(cam1, Xi, u, v, w1);
    const double e1 = sqrt((u - x1) * (u - x1) + (v - y1) * (v - y1));
    for (int c = 0; c < 3; ++c) a.X[i * a.sX.xi + c * a.sX.xc] = Xi[c];
    a.reproj[i] = 0.5 * (e0 + e1);
    a.valid[i] = (uint8_t)(w0 > 0.0 && w1 > 0.0);
}

__global__ __launch_bounds__(256) void k_angle(const double* X, vo_strides sX, int n, const double* pose0,
                                               const double* pose1, int pose_col_major, double* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double p0[12], p1[12], Xi[3];
    load43(pose0, pose_col_major, p0);
    load43(pose1, pose_col_major, p1);
    for (int c = 0; c < 3; ++c) Xi[c] = X[i * sX.xi + c * sX.xc];
    out[i] = angle_deg(Xi, p0 + 9, p1 + 9);
}

__global__ __launch_bounds__(256) void k_in_front(const double* cam_, int cam_col_major, const double* X, vo_strides sX,
                                                  int n, uint8_t* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double cam[12];
    load43(cam_, cam_col_major, cam);
    const double x = X[i * sX.xi], y = X[i * sX.xi + sX.xc], z = X[i * sX.xi + 2 * sX.xc];
    const double w = ((x * cam[2] + y * cam[5]) + z * cam[8]) + cam[11];
    out[i] = (uint8_t)(w > 0.0);
}

__global__ __launch_bounds__(256) void k_reproj(const double* xy, vo_strides sxy, const double* X, vo_strides sX, int n,
                                                const double* pose_, int pose_col_major, const double* K_, double* err,
                                                const int32_t* n_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (n_dev ? min(n, *n_dev) : n)) return;
    double pose[12], K[9], cam[12], Xi[3];
    load43(pose_, pose_col_major, pose);
    for (int k = 0; k < 9; ++k) K[k] = K_[k];
    camera_matrix(pose, K, cam);
    for (int c = 0; c < 3; ++c) Xi[c] = X[i * sX.xi + c * sX.xc];
    double u, v, w;
    project(cam, Xi, u, v, w);
    const double du = xy[i * sxy.xi] - u, dv = xy[i * sxy.xi + sxy.xc] - v;
    err[i] = sqrt(du * du + dv * dv);
}

} // namespace

int vo_launch_triangulate(vo_ctx* ctx, const double* d_xy0, const double* d_xy1, vo_strides sxy,
                          const double* d_poses0, int pose_col_major, const int32_t* d_pose_idx, int C, int P,
                          const double* d_pose1, const double* d_K, double* d_X, vo_strides sX,
                          double* d_reproj, double* d_angle, uint8_t* d_valid, const int32_t* d_C) {
    if (C <= 0) return VO_OK;
    TriArgs a;
    a.xy0 = d_xy0; a.xy1 = d_xy1; a.sxy = sxy; a.poses0 = d_poses0; a.pose_col_major = pose_col_major;
    a.pose_idx = d_pose_idx; a.C = C; a.P = P; a.pose1 = d_pose1; a.K = d_K; a.X = d_X; a.sX = sX;
    a.reproj = d_reproj; a.angle = d_angle; a.valid = d_valid; a.C_dev = d_C;
    hipLaunchKernelGGL(k_triangulate, dim3((C + 127) / 128), dim3(256), 0, ctx->stream, a);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_triangulate_cam(vo_ctx* ctx, const double* d_xy0, const double* d_xy1, vo_strides sxy, const double* d_cam0,
                              const double* d_cam1, int cam_col_major, int n, double* d_X, vo_strides sX, double* d_reproj,
                              uint8_t* d_valid) {
    if (n <= 0) return VO_OK;
    TriCamArgs a;
    a.xy0 = d_xy0; a.xy1 = d_xy1; a.sxy = sxy; a.cam0 = d_cam0; a.cam1 = d_cam1; a.cam_col_major = cam_col_major; a.n = n;
    a.X = d_X; a.sX = sX; a.reproj = d_reproj; a.valid = d_valid;
    hipLaunchKernelGGL(k_triangulate_cam, dim3((n + 127) / 128), dim3(256), 0, ctx->stream, a);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_angle(vo_ctx* ctx, const double* d_X, vo_strides sX, int n, const double* d_pose0,
                    const double* d_pose1, int pose_col_major, double* d_angle) {
    if (n <= 0) return VO_OK;
    hipLaunchKernelGGL(k_angle, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_X, sX, n, d_pose0, d_pose1,
                       pose_col_major, d_angle);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_in_front(vo_ctx* ctx, const double* d_cam, int cam_col_major, const double* d_X, vo_strides sX,
                       int n, uint8_t* d_valid) {
    if (n <= 0) return VO_OK;
    hipLaunchKernelGGL(k_in_front, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_cam, cam_col_major, d_X, sX, n, d_valid);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_reproj(vo_ctx* ctx, const double* d_xy, vo_strides sxy, const double* d_X, vo_strides sX, int n,
                     const double* d_pose, int pose_col_major, const double* d_K, double* d_err, const int32_t* d_n) {
    if (n <= 0) return VO_OK;
    hipLaunchKernelGGL(k_reproj, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_xy, sxy, d_X, sX, n, d_pose,
                       pose_col_major, d_K, d_err, d_n);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
