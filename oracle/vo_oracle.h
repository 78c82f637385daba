/*
 * vo_oracle.h -- CPU parity oracle for the VO per-frame hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (visual_odometry_amd/, libvo_hip.so) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (giuliano-97/visual_odometry, pure MATLAB) ships no golden
 * vectors, numeric assertions or datasets, and its arithmetic lives in closed-source MathWorks
 * toolboxes (Computer Vision Toolbox, no pinned version).  MATLAB/Octave are
 * absent from the build image, so nothing of the reference can be executed here.  Functions that
 * restate OPEN reference code (old/p3p.m, old/solveQuartic.m, old/ransacLocalization.m,
 * old/harris.m, old/selectKeypoints.m, old/linearTriangulation.m, utils/calculateAngleDeg.m,
 * modules/isInFrontOfCamera.m) follow it line by line and cite it; toolbox-internal steps are
 * DEFINED here (normative definitions N1..N10 below; N11 rgb2gray and N12 JPEG arithmetic of the
 * frame ingest are in oracle/ingest.py) from the public documentation and the
 * Bouguet / Shi-Tomasi / Kneip / Torr-Zisserman papers.  The oracle is pinned by analytic
 * known-answer tests (tests/test_oracle_*.py) and by the fixtures under tests/golden/ that it
 * emitted itself (tests/golden/make_golden.py).
 *
 * All arithmetic is IEEE double, compiled with -ffp-contract=off (the compiler fuses nothing; the
 * only fused operations are the explicit fma() calls of the N9 scoring), so that the integer /
 * double decision paths (selection indices, inlier sets) can be reproduced bit-exactly by the HIP
 * side, which uses the same operation order.
 *
 * ---------------------------------------------------------------------------------------------
 * Internal image convention: an image is `nu` x `nv` uint8, u = FAST (contiguous) axis, v = slow
 * axis, pixel (u,v) at img[v*nu+u], coordinates 0-based at pixel centres.  A MATLAB H x W
 * column-major image is (nu,nv)=(H,W) with u=y-1, v=x-1; a numpy H x W row-major image is
 * (nu,nv)=(W,H) with u=x, v=y.  Nothing is ever transposed; callers map coordinates.
 *
 * N1 pyramid     level 0 = (double)uint8.  level l+1(i,j) = sum_a sum_b w_a w_b L_l(clamp(2i+a-2),
 *                clamp(2j+b-2)), w = [1 4 6 4 1]/16, replicate border, size ceil(n/2).
 *                Horizontal (u) pass first, then vertical.
 * N2 gradients   Sobel/8, replicate border: gu = ([1 2 1]^T along v) x ([-1 0 1] along u) / 8.
 * N3 bilinear    sample(I,u,v): iu=floor(u), fu=u-iu, taps clamp(iu), clamp(iu+1) (replicate).
 * N4 LK          per point, coarse->fine, forward-additive with fixed template gradient
 *                (Bouguet 2001): E = bilinear samples of image A on the (wu+2) x (wv+2) window
 *                centred at p/2^l; template T = interior of E; Tu,Tv = Sobel/8 (N2 operator) of E,
 *                i.e. the gradient of the sub-pixel-sampled patch; G = sum [Tu^2 TuTv; TuTv Tv^2];
 *                level is "flat" if mineig(G)/npx < VO_MIN_EIG (then no update at that level; at
 *                level 0 the point is invalid).  Iterate k < max_iters: J = samples of image B at
 *                p/2^l + g + d; b = sum (T-J)[Tu Tv]; delta = G^-1 b; d += delta; stop when
 *                |delta|^2 < eps^2.  Lost if (p/2^l+g+d)*2^l leaves [0,nu-1]x[0,nv-1].
 *                g <- 2(g+d) between levels.
 * N5 validity    forward track A->B from p gives q; backward track B->A from q gives p';
 *                valid = !lost_fwd & !lost_bwd & !flat0 & |p-p'| <= max_bidir_err (if >= 0... a
 *                negative max_bidir_err disables the backward pass).
 *                score = clamp(1 - sqrt(SSD(T - J(q))/npx)/255, 0, 1) at level 0, 0 if lost.
 * N6 response    MINEIGEN: d = central difference (I(u+1)-I(u-1), replicate), integer; products
 *                a=du^2 b=du*dv c=dv^2; separable Gaussian with integer weights
 *                w_k = lround(4096*exp(-k^2/(2 s^2))/sum), s = filter_size/3, replicate border,
 *                int64 sums A,B,C; metric = 0.5*((A+C) - sqrt((A-C)^2 + 4 B^2)) * 2^-26.
 *                HARRIS: old/harris.m verbatim in double (Sobel 'valid', box sum 'valid',
 *                det - kappa*trace^2, negatives -> 0, zero border 1+pr).
 * N7 detection   thr = min_quality*max(metric); corner iff metric > thr, not on the 1-px border,
 *                and (metric, -linear index) is strictly greatest among its 8 neighbours.
 *                Sub-pixel: off = clamp(-0.5*(r+ - r-)/(r+ - 2r + r-), +-0.5) if curvature < 0.
 *                Corner list order = ascending linear index v*nu+u.
 * N8 filtering   drop corner if any existing point has sqrtf(dx*dx+dy*dy) < min_distance in
 *                single precision (modules/selectCandidateKeypoints.m:23-29, strict <);
 *                selectUniform: rounds of "strongest per grid cell" (ties: lower index) with
 *                h=max(floor(sqrt(n/aspect)),1), w=max(floor(h*aspect),1) until n points.
 * N9 P3P-RANSAC  Philox4x32-10 sample stream; Kneip P3P + Ferrari quartic (complex).  Scoring in
 *                double: Pm = K [R_cw | t_cw] per pose; (u~, v~, z) = Pm [X; 1] by fma chains;
 *                inlier iff (u*z - u~)^2 + (v*z - v~)^2 < thr^2 * z^2 (the homogeneous form of
 *                |p - proj|^2 < thr^2, old/ransacLocalization.m:96-99; z = 0 is never an inlier);
 *                MSAC cost = sum over inliers of floor(e2 * 2^20), e2 = ez / z^2, plus
 *                floor(thr^2 * 2^20) per outlier; see p3p.c.
 * N10 essential   bootstrap's estimateEssentialMatrix: MSAC around the normalised eight-point algorithm,
 *                Sampson distance in pixels^2, Philox samples of 8; see essential.c.
 * N10b            the same around Nister's five-point solver (samples of 5, up to ten matrices each); see five_point.c.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_MIN_EIG 1e-3
#define VO_MAX_LEVELS 8

typedef struct {
    int levels;          /* NumPyramidLevels           (modules/KLTTracker.m:11) */
    double max_bidir;    /* MaxBidirectionalError, <0: no backward pass (:12) */
    int wu, wv;          /* BlockSize along u and v, odd (:13) */
    int max_iters;       /* MaxIterations              (:14) */
    double eps;          /* convergence threshold in pixels (N4) */
    int threads;         /* OpenMP threads for the cpu_baseline timing, 1 = scalar */
} vo_or_klt_opts;

/* pyramid geometry: returns total number of pixels, fills per-level sizes and offsets */
long vo_or_pyramid_layout(int nu, int nv, int levels, int* nus, int* nvs, long* offs);
void vo_or_build_pyramid(const uint8_t* img, int nu, int nv, int levels, double* pyr);
void vo_or_gradients(const double* I, int nu, int nv, double* gu, double* gv);

/* modules/KLTTracker.m:22-41 track(); pts are internal (u,v) 0-based, SoA.  iters_out (optional)
 * receives the total number of LK iterations executed per point (fwd+bwd, all levels). */
int vo_or_klt_track(const uint8_t* img1, const uint8_t* img2, int nu, int nv,
                    const double* pu, const double* pv, int n, const vo_or_klt_opts* o,
                    double* qu, double* qv, uint8_t* valid, double* score,
                    double* fb_err, int* iters_out);

/* ---- candidate selection (modules/selectCandidateKeypoints.m:1-36) ---- */
enum { VO_SCORER_MINEIGEN = 0, VO_SCORER_HARRIS = 1 };
void vo_or_gauss_weights(int filter_size, int32_t* w);
void vo_or_corner_response(const uint8_t* img, int nu, int nv, int scorer, int filter_size,
                           double kappa, double* resp);
/* N7: returns number of corners; lin_idx ascending, loc (float) in internal coords, metric */
int vo_or_detect_corners(const double* resp, int nu, int nv, double min_quality, int max_out,
                         int32_t* lin_idx, float* loc_u, float* loc_v, double* metric);
/* N8 distance filter: keep[i]=1 if no existing point closer than min_distance */
void vo_or_distance_filter(const float* cu, const float* cv, int m, const float* eu,
                           const float* ev, int n_exist, float min_distance, uint8_t* keep);
/* N8 selectUniform on the kept corners; sel[i]=1 if selected; cells_* from image aspect */
int vo_or_select_uniform(const float* cu, const float* cv, const double* metric,
                         const uint8_t* keep, int m, int n_keep, int nu, int nv,
                         int aspect_u_over_v_is_h_over_w, uint8_t* sel);
/* full a3 pipeline; out arrays sized `keep`; returns M */
/* threads for the independent loops of selection and triangulation (default 1; results do not depend on it) */
void vo_or_set_threads(int n);
int vo_or_select_candidates(const uint8_t* img, int nu, int nv, const float* eu, const float* ev,
                            int n_exist, float min_distance, double min_quality,
                            int filter_size, int keep, int scorer, double kappa, int col_major,
                            float* out_u, float* out_v, int32_t* out_lin, double* out_metric);
/* old/selectKeypoints.m:5-14 greedy top-N with box suppression; out rows/cols 0-based */
void vo_or_select_keypoints_greedy(const double* scores, int nu, int nv, int num, int r,
                                   int32_t* out_u, int32_t* out_v);

/* ---- geometry (utils/calculateAngleDeg.m, modules/isInFrontOfCamera.m, triangulate) ---- */
/* pose = 4x3 row-major [R_WC; t_WC] (modules/VisualOdometry.m:318); K 3x3 row-major (not K^T) */
void vo_or_camera_matrix(const double* pose43, const double* K, double* cam43);
double vo_or_angle_deg(const double* X, const double* pose0, const double* pose1);
void vo_or_in_front(const double* cam43, const double* X, int n, uint8_t* valid);
void vo_or_triangulate(const double* xy0, const double* xy1, const double* poses0 /* n x 12 */,
                       const double* pose1, const double* K, int n, double* X, double* reproj,
                       double* angle_deg, uint8_t* valid);
void vo_or_triangulate_cam(const double* xy0, const double* xy1, const double* cam0 /* 4x3 row-major */, const double* cam1,
                           int n, double* X, double* reproj, uint8_t* valid);
void vo_or_reproj_error(const double* xy, const double* X, int n, const double* pose43,
                        const double* K, double* err);

/* ---- P3P-RANSAC (old/p3p.m, old/solveQuartic.m, old/ransacLocalization.m) ---- */
void vo_or_solve_quartic(const double* factors5, double* roots_re_im /* 4x2 */);
void vo_or_p3p(const double* P /* 3 pts x 3, row per point */, const double* f /* 3 bearings x 3 */,
               double* poses /* 4 x 12: C(3) then R row-major(9), R cam->world */);
void vo_or_philox_sample3(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx3);

typedef struct {
    int num_hyp;          /* MaxNumTrials */
    double max_reproj;    /* MaxReprojectionError (pixels) */
    double confidence;    /* in (0,1): adaptive stop; <=0: evaluate all hypotheses */
    uint64_t seed;
    int mode;             /* 0 = inlier count (old/ransacLocalization.m), 1 = MSAC cost */
    int min_inliers;      /* old/ransacLocalization.m:102,107-108 */
    int threads;
} vo_or_ransac_opts;

/* returns status 0 ok / 1 fewer than 4 points / 2 not enough inliers.  R = R_cw (row-major,
 * x_c = R X + t), t = t_cw.  sample_idx optional H x 3 (0-based).  hyp_count/hyp_cost optional
 * outputs (H x 4) for parity inspection; trials_used = hypotheses consumed before the adaptive
 * stop. */
int vo_or_p3p_ransac(const double* xy, const double* X, int m, const double* K,
                     const vo_or_ransac_opts* o, const int32_t* sample_idx, double* R, double* t,
                     uint8_t* inliers, int* best_hyp, int* best_sol, int* trials_used,
                     int32_t* hyp_count, uint64_t* hyp_cost, double* hyp_poses /* H x 4 x 12 */);

/* N10 (essential.c): bootstrap's estimateEssentialMatrix, modules/bootstrap.m:76-79.  Points are
 * m x 2 pixel coordinates in K's origin; o->max_reproj is MaxDistance (Sampson, pixels^2),
 * o->min_inliers the least count a winning hypothesis needs.  Status 0 ok / 1 fewer than 8 points /
 * 2 no hypothesis qualified.  sample_idx optional H x 8.  E satisfies x1' E x0 = 0 on K^-1-normalised
 * points and has singular values (1, 1, 0). */
void vo_or_philox_sample8(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx8);
int vo_or_eight_point(const double* x0, const double* y0, const double* x1, const double* y1, double* E);
int vo_or_essential_project(const double* E0, double* E);
void vo_or_fundamental_from_essential(const double* E, const double* K, double* F);
int vo_or_essential_ransac(const double* p0, const double* p1, int m, const double* K,
                           const vo_or_ransac_opts* o, const int32_t* sample_idx, double* E,
                           uint8_t* inliers, int* best_hyp, int* trials_used, int32_t* hyp_count,
                           uint64_t* hyp_cost, double* hyp_E /* H x 9 */);
/* N10b (five_point.c): the same step around Nister's five-point solver.  Es: up to 10 matrices per sample, ascending
 * in the root; the hyp_* arrays of the loop have 10 slots per sample and best_hyp is a slot index (10 h + s). */
void vo_or_philox_sample5(uint64_t seed, uint32_t hyp, uint32_t m, int32_t* idx5);
int vo_or_five_point(const double* x0, const double* y0, const double* x1, const double* y1, double* Es /* 10 x 9 */);
int vo_or_essential_ransac5(const double* p0, const double* p1, int m, const double* K,
                            const vo_or_ransac_opts* o, const int32_t* sample_idx, double* E,
                            uint8_t* inliers, int* best_hyp, int* trials_used, int32_t* hyp_count,
                            uint64_t* hyp_cost, double* hyp_E);

#ifdef __cplusplus
}
#endif
#endif
