/* oracle/geom.c -- CPU restatement of the per-candidate geometry (TEST INFRASTRUCTURE).
 *
 * Reference: modules/VisualOdometry.m:48-61 (reprojection error), :81-101 (camera matrices +
 * two-view triangulate), :106,:113 (gates); utils/calculateAngleDeg.m:7-10;
 * modules/isInFrontOfCamera.m:8-11; old/linearTriangulation.m:28-39 (DLT via SVD).
 *
 * Conventions: pose = 4x3 [R_WC; t_WC], MATLAB row-vector convention
 * x_cam = (X - t) * R^T.  Extrinsics Rext = R_WC^T, text = -t_WC * R_WC^T
 * (cameraPoseToExtrinsics); camera matrix (4x3) = [Rext; text] * K^T (cameraMatrix);
 * projection [u v w] = [X 1] * camMat (worldToImage).  K here is the usual 3x3 (row-major), i.e.
 * the transpose of cameraParams.IntrinsicMatrix (utils/dataLoaderKitti.m:50-53).
 * All 4x3 matrices are stored row-major (12 doubles).
 */
#include "vo_oracle.h"
#include <math.h>
#include <string.h>

void vo_or_camera_matrix(const double* pose, const double* K, double* cam) {
    /* Rext = R^T  (Rext[i][j] = R[j][i]);  text[j] = -sum_i t[i] * R^T[i][j] = -sum_i t[i]*R[j][i] */
    double E[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[i * 3 + j] = pose[j * 3 + i];
    for (int j = 0; j < 3; ++j)
        E[9 + j] = -((pose[9] * pose[j * 3 + 0] + pose[10] * pose[j * 3 + 1]) + pose[11] * pose[j * 3 + 2]);
    /* cam = E(4x3) * K^T : cam[i][j] = sum_k E[i][k] * K[j][k] */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            cam[i * 3 + j] = (E[i * 3 + 0] * K[j * 3 + 0] + E[i * 3 + 1] * K[j * 3 + 1]) + E[i * 3 + 2] * K[j * 3 + 2];
}

/* utils/calculateAngleDeg.m:7-10; cos clamped to [-1,1] (MATLAB would go complex) */
double vo_or_angle_deg(const double* X, const double* p0, const double* p1) {
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = X[i] - p0[9 + i]; b[i] = X[i] - p1[9 + i]; }
    double dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    double na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    double nb = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
    double c = dot / (na * nb);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    return acos(c) * (180.0 / 3.14159265358979323846);
}

/* modules/isInFrontOfCamera.m:8-11 */
void vo_or_in_front(const double* cam, const double* X, int n, uint8_t* valid) {
    for (int i = 0; i < n; ++i) {
        double w = ((X[i * 3 + 0] * cam[2] + X[i * 3 + 1] * cam[5]) + X[i * 3 + 2] * cam[8]) + cam[11];
        valid[i] = (uint8_t)(w > 0.0);
    }
}

static void project(const double* cam, const double* X, double* u, double* v, double* w) {
    double pu = ((X[0] * cam[0] + X[1] * cam[3]) + X[2] * cam[6]) + cam[9];
    double pv = ((X[0] * cam[1] + X[1] * cam[4]) + X[2] * cam[7]) + cam[10];
    double pw = ((X[0] * cam[2] + X[1] * cam[5]) + X[2] * cam[8]) + cam[11];
    *u = pu / pw; *v = pv / pw; *w = pw;
}

/* one rotation of columns p, q of A (and of V) that makes them orthogonal; returns 1 unless it was the
 * identity up to rounding (c == 1.0).  gamma == 0: c = 1, s = 0, applied like any other. */
static int jacobi_rotate(double* A, double* V, int p, int q) {
    double alpha = 0, beta = 0, gamma = 0;
    for (int k = 0; k < 4; ++k) {
        alpha += A[k * 4 + p] * A[k * 4 + p];
        beta += A[k * 4 + q] * A[k * 4 + q];
        gamma += A[k * 4 + p] * A[k * 4 + q];
    }
    double zeta = (beta - alpha) / (2.0 * gamma);
    double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    if (gamma == 0.0) { c = 1.0; s = 0.0; }
    for (int k = 0; k < 4; ++k) {
        double ap = A[k * 4 + p], aq = A[k * 4 + q];
        A[k * 4 + p] = c * ap - s * aq; A[k * 4 + q] = s * ap + c * aq;
        double vp = V[k * 4 + p], vq = V[k * 4 + q];
        V[k * 4 + p] = c * vp - s * vq; V[k * 4 + q] = s * vp + c * vq;
    }
    return c != 1.0;
}

/* smallest right-singular vector of a 4x4 matrix by one-sided (Hestenes) Jacobi.  A sweep is three rounds of
 * two disjoint column pairs, (0,1)(2,3) / (0,2)(1,3) / (0,3)(1,2) (the two rotations of a round touch different
 * columns: the device runs them side by side); the iteration ends with the first sweep all of whose rotations
 * were the identity up to rounding, after 12 sweeps at the latest.  A is row-major and is overwritten. */
static void svd4_last(double* A, double* vout) {
    static const int PAIRS[6][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {0, 3}, {1, 2}};
    double V[16];
    memset(V, 0, sizeof V);
    for (int i = 0; i < 4; ++i) V[i * 4 + i] = 1.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < 6; ++r) rotated |= jacobi_rotate(A, V, PAIRS[r][0], PAIRS[r][1]);
        if (!rotated) break;
    }
    int best = 0; double bn = INFINITY;
    for (int j = 0; j < 4; ++j) {
        double nn = 0;
        for (int k = 0; k < 4; ++k) nn += A[k * 4 + j] * A[k * 4 + j];
        if (nn < bn) { bn = nn; best = j; }
    }
    for (int k = 0; k < 4; ++k) vout[k] = V[k * 4 + best];
}

/* modules/VisualOdometry.m:93-113 per candidate (batched): DLT rows x*P3 - P1, y*P3 - P2 per view
 * (the 4x4 form of old/linearTriangulation.m:29-36), reprojection error = mean over the two
 * views, validity = positive depth in both views, angle per utils/calculateAngleDeg.m. */
void vo_or_triangulate(const double* xy0, const double* xy1, const double* poses0, const double* pose1,
                       const double* K, int n, double* X, double* reproj, double* angle, uint8_t* valid) {
    double cam1[12];
    vo_or_camera_matrix(pose1, K, cam1);
    extern int vo_or_threads;   /* select.c */
#pragma omp parallel for num_threads(vo_or_threads) schedule(static)
    for (int i = 0; i < n; ++i) {
        double cam0[12], A[16], h[4];
        vo_or_camera_matrix(poses0 + 12 * i, K, cam0);
        const double x0 = xy0[2 * i], y0 = xy0[2 * i + 1], x1 = xy1[2 * i], y1 = xy1[2 * i + 1];
        for (int k = 0; k < 4; ++k) { /* P = cam^T: P1[k] = cam[k][0], P2[k] = cam[k][1], P3[k] = cam[k][2] */
            A[0 * 4 + k] = x0 * cam0[k * 3 + 2] - cam0[k * 3 + 0];
            A[1 * 4 + k] = y0 * cam0[k * 3 + 2] - cam0[k * 3 + 1];
            A[2 * 4 + k] = x1 * cam1[k * 3 + 2] - cam1[k * 3 + 0];
            A[3 * 4 + k] = y1 * cam1[k * 3 + 2] - cam1[k * 3 + 1];
        }
        svd4_last(A, h);
        double Xi[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
        double u, v, w0, w1, e0, e1;
        project(cam0, Xi, &u, &v, &w0);
        e0 = sqrt((u - x0) * (u - x0) + (v - y0) * (v - y0));
        project(cam1, Xi, &u, &v, &w1);
        e1 = sqrt((u - x1) * (u - x1) + (v - y1) * (v - y1));
        X[3 * i] = Xi[0]; X[3 * i + 1] = Xi[1]; X[3 * i + 2] = Xi[2];
        reproj[i] = 0.5 * (e0 + e1);
        valid[i] = (uint8_t)(w0 > 0.0 && w1 > 0.0);
        angle[i] = vo_or_angle_deg(Xi, poses0 + 12 * i, pose1);
    }
}

/* the toolbox call itself, `triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2)` as the reference makes
 * it (modules/VisualOdometry.m:97-101, modules/candidateTriangulation.m:47-51, processFrame.m:55-59): the same DLT,
 * error and validity as above with the two 4x3 camera matrices given (row-major), no angle. */
void vo_or_triangulate_cam(const double* xy0, const double* xy1, const double* cam0, const double* cam1, int n,
                           double* X, double* reproj, uint8_t* valid) {
    for (int i = 0; i < n; ++i) {
        double A[16], h[4];
        const double x0 = xy0[2 * i], y0 = xy0[2 * i + 1], x1 = xy1[2 * i], y1 = xy1[2 * i + 1];
        for (int k = 0; k < 4; ++k) {
            A[0 * 4 + k] = x0 * cam0[k * 3 + 2] - cam0[k * 3 + 0];
            A[1 * 4 + k] = y0 * cam0[k * 3 + 2] - cam0[k * 3 + 1];
            A[2 * 4 + k] = x1 * cam1[k * 3 + 2] - cam1[k * 3 + 0];
            A[3 * 4 + k] = y1 * cam1[k * 3 + 2] - cam1[k * 3 + 1];
        }
        svd4_last(A, h);
        double Xi[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
        double u, v, w0, w1, e0, e1;
        project(cam0, Xi, &u, &v, &w0);
        e0 = sqrt((u - x0) * (u - x0) + (v - y0) * (v - y0));
        project(cam1, Xi, &u, &v, &w1);
        e1 = sqrt((u - x1) * (u - x1) + (v - y1) * (v - y1));
        X[3 * i] = Xi[0]; X[3 * i + 1] = Xi[1]; X[3 * i + 2] = Xi[2];
        reproj[i] = 0.5 * (e0 + e1);
        valid[i] = (uint8_t)(w0 > 0.0 && w1 > 0.0);
    }
}

/* modules/VisualOdometry.m:48-61 */
void vo_or_reproj_error(const double* xy, const double* X, int n, const double* pose, const double* K, double* err) {
    double cam[12];
    vo_or_camera_matrix(pose, K, cam);
    for (int i = 0; i < n; ++i) {
        double u, v, w;
        project(cam, X + 3 * i, &u, &v, &w);
        double du = xy[2 * i] - u, dv = xy[2 * i + 1] - v;
        err[i] = sqrt(du * du + dv * dv);
    }
}
