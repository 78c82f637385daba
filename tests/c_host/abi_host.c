/* abi_host.c -- the C ABI from a plain C host: no Python, no torch, nothing but libvo_hip.so and the HIP runtime it
 * links -- what a MEX gateway's process looks like (mex/).  Calls the host-pointer entry points
 * with MATLAB's layout (column-major, 1-based: VO_MATLAB of mex/vo_mex_common.h) on synthetic inputs and checks them
 * against the oracle (TEST INFRASTRUCTURE, linked here as the checker only): tracks within 0.01 px, selection
 * indices bit-exact, triangulation to 1e-12 relative.  Built by `make -C tests/c_host` (gcc; __graft_entry__.build()),
 * run by tests/test_gpu_c_host.py.  Exit status 0 = all checks passed; one summary line per check on stdout. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vo_hip.h"

/* the oracle's entry points used here, restated (oracle/vo_oracle.h and include/vo_hip.h both define VO_SCORER_*) */
typedef struct { int levels; double max_bidir; int wu, wv; int max_iters; double eps; int threads; } vo_or_klt_opts;
int vo_or_klt_track(const uint8_t* img1, const uint8_t* img2, int nu, int nv, const double* pu, const double* pv, int n,
                    const vo_or_klt_opts* o, double* qu, double* qv, uint8_t* valid, double* score, double* fb_err, int* iters_out);
int vo_or_select_candidates(const uint8_t* img, int nu, int nv, const float* eu, const float* ev, int n_exist, float min_distance,
                            double min_quality, int filter_size, int keep, int scorer, double kappa, int col_major, float* out_u,
                            float* out_v, int32_t* out_lin, double* out_metric);
void vo_or_camera_matrix(const double* pose43, const double* K, double* cam43);
void vo_or_triangulate_cam(const double* xy0, const double* xy1, const double* cam0, const double* cam1, int n, double* X,
                           double* reproj, uint8_t* valid);

static uint32_t lcg_state = 12345u;
static double lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (lcg_state >> 8) / 16777216.0; }

static double scene(double x, double y) {   /* a smooth texture with structure at several scales */
    return 128.0 + 34.0 * sin(0.11 * x + 0.3) * sin(0.07 * y) + 28.0 * sin(0.23 * x + 0.17 * y + 1.0) +
           24.0 * sin(0.05 * x - 0.31 * y + 2.0) + 18.0 * sin(0.41 * x + 0.13 * y) + 12.0 * sin(0.63 * y - 0.37 * x + 0.5);
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != VO_OK) { \
    fprintf(stderr, "%s failed: %s [%s]\n", #call, vo_strerror(rc_), ctx ? vo_ctx_last_error(ctx) : ""); return 2; } } while (0)

int main(void) {
    enum { H = 240, W = 320, N = 600, KEEP = 150, NT = 400 };
    const vo_layout matlab = {1, 1};
    vo_ctx* ctx = NULL;
    CHECK(vo_ctx_create(0, &ctx));
    int failures = 0;

    /* ---- images, MATLAB memory: pixel (x, y) 1-based at (x - 1) * H + (y - 1) ---- */
    uint8_t* img1 = malloc((size_t)H * W); uint8_t* img2 = malloc((size_t)H * W);
    const double sx = 2.3, sy = -1.4;
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y) {
            img1[(size_t)x * H + y] = (uint8_t)lrint(scene(x, y));
            img2[(size_t)x * H + y] = (uint8_t)lrint(scene(x - sx, y - sy));   /* content moves by (+sx, +sy) */
        }

    /* ---- KLTTracker.track ---- */
    {
        double* pts = malloc(sizeof(double) * 2 * N); double* out = malloc(sizeof(double) * 2 * N); double* sc = malloc(sizeof(double) * N);
        uint8_t* valid = malloc(N);
        double* pu = malloc(sizeof(double) * N); double* pv = malloc(sizeof(double) * N);
        for (int i = 0; i < N; ++i) {
            const double x = 14.0 + lcg() * (W - 28.0), y = 14.0 + lcg() * (H - 28.0);   /* 1-based */
            pts[i] = x; pts[N + i] = y;              /* N x 2 column-major */
            pu[i] = y - 1.0; pv[i] = x - 1.0;        /* oracle: u = fast axis = y - 1, v = x - 1 */
        }
        const vo_klt_opts o = {4, 2.0, 21, 21, 30, 0.01};
        CHECK(vo_klt_track(ctx, img1, img2, H, W, matlab, pts, N, 1, &o, out, valid, sc));
        const vo_or_klt_opts oo = {4, 2.0, 21, 21, 30, 0.01, 1};
        double* qu = malloc(sizeof(double) * N); double* qv = malloc(sizeof(double) * N); double* osc = malloc(sizeof(double) * N);
        double* fb = malloc(sizeof(double) * N); uint8_t* ovalid = malloc(N);
        vo_or_klt_track(img1, img2, H, W, pu, pv, N, &oo, qu, qv, ovalid, osc, fb, NULL);
        int both = 0, flips = 0; double worst = 0, worst_truth = 0, worst_score = 0;
        for (int i = 0; i < N; ++i) {
            if (valid[i] != ovalid[i]) { if (!(fabs(fb[i] - 2.0) < 1e-3)) ++flips; continue; }
            if (!valid[i]) continue;
            ++both;
            const double du = (out[N + i] - 1.0) - qu[i], dv = (out[i] - 1.0) - qv[i];
            worst = fmax(worst, hypot(du, dv));
            worst_truth = fmax(worst_truth, hypot(out[i] - (pts[i] + sx), out[N + i] - (pts[N + i] + sy)));
            worst_score = fmax(worst_score, fabs(sc[i] - osc[i]));
        }
        const int ok = flips == 0 && both > N * 9 / 10 && worst < 0.01 && worst_score < 1e-3;
        printf("klt_track: %d of %d valid on both sides, %d validity flips, worst |gpu - oracle| %.2e px, worst |gpu - truth| %.3f px, worst score difference %.1e: %s\n",
               both, N, flips, worst, worst_truth, worst_score, ok ? "ok" : "FAILED");
        failures += !ok;
        /* N = 0: VO_OK and nothing touched (modules/KLTTracker.m:33-36) */
        CHECK(vo_klt_track(ctx, img1, img2, H, W, matlab, pts, 0, 1, &o, out, valid, sc));

        /* ---- selectCandidateKeypoints against the tracked points ---- */
        float* ex = malloc(sizeof(float) * 2 * N); float* eu = malloc(sizeof(float) * N); float* ev = malloc(sizeof(float) * N);
        for (int i = 0; i < N; ++i) {
            ex[i] = (float)pts[i]; ex[N + i] = (float)pts[N + i];
            eu[i] = ex[N + i] - 1.0f; ev[i] = ex[i] - 1.0f;
        }
        const vo_sel_opts so = {7.0, 0.001, 5, KEEP, VO_SCORER_MINEIGEN, 0.08};
        float oxy[2 * KEEP]; int32_t olin[KEEP]; double omet[KEEP]; int M = -1;
        CHECK(vo_select_candidates(ctx, img2, H, W, matlab, ex, N, 0, &so, oxy, olin, omet, &M));
        float ru[KEEP], rv[KEEP]; int32_t rlin[KEEP]; double rmet[KEEP];
        const int Mo = vo_or_select_candidates(img2, H, W, eu, ev, N, 7.0f, 0.001, 5, KEEP, VO_SCORER_MINEIGEN, 0.08, 1, ru, rv, rlin, rmet);
        /* indices and metrics bit for bit; the sub-pixel locations come back 1-based (float + 1.0f rounds: an ulp at 300) */
        int same = M == Mo;
        double worst_loc = 0;
        for (int i = 0; same && i < M; ++i) {
            same = olin[i] == rlin[i] && omet[i] == rmet[i];
            worst_loc = fmax(worst_loc, fmax(fabs((double)oxy[M + i] - 1.0 - ru[i]), fabs((double)oxy[i] - 1.0 - rv[i])));
        }
        const int sel_ok = same && M > 20 && worst_loc < 5e-5;
        printf("select_candidates: %d corners (oracle %d), indices and metrics %s, worst location difference %.1e px: %s\n", M, Mo,
               same ? "bit-equal" : "DIFFER", worst_loc, sel_ok ? "ok" : "FAILED");
        failures += !sel_ok;
    }

    /* ---- triangulate(p1, p2, camMatrix1, camMatrix2): two views of NT points ---- */
    {
        const double K[9] = {718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1};
        const double a = 2.0 * M_PI / 180.0;
        const double pose0[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        const double pose1[12] = {cos(a), 0, -sin(a), 0, 1, 0, sin(a), 0, cos(a), 0.9, 0.05, 0.3};   /* [R_wc; t_wc] row-major */
        double c0[12], c1[12], c0m[12], c1m[12];
        vo_or_camera_matrix(pose0, K, c0); vo_or_camera_matrix(pose1, K, c1);
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) { c0m[c * 4 + r] = c0[r * 3 + c]; c1m[c * 4 + r] = c1[r * 3 + c]; }
        double* p0 = malloc(sizeof(double) * 2 * NT); double* p1 = malloc(sizeof(double) * 2 * NT);
        double* r0 = malloc(sizeof(double) * 2 * NT); double* r1 = malloc(sizeof(double) * 2 * NT);
        for (int i = 0; i < NT; ++i) {
            const double X[3] = {-12.0 + 24.0 * lcg(), -3.0 + 6.0 * lcg(), 6.0 + 50.0 * lcg()};
            const double* cams[2] = {c0, c1}; double* rows[2] = {r0, r1}; double* cols[2] = {p0, p1};
            for (int k = 0; k < 2; ++k) {
                const double* c = cams[k];
                const double w = X[0] * c[2] + X[1] * c[5] + X[2] * c[8] + c[11];
                const double u = (X[0] * c[0] + X[1] * c[3] + X[2] * c[6] + c[9]) / w + (lcg() - 0.5);
                const double v = (X[0] * c[1] + X[1] * c[4] + X[2] * c[7] + c[10]) / w + (lcg() - 0.5);
                rows[k][2 * i] = u; rows[k][2 * i + 1] = v; cols[k][i] = u; cols[k][NT + i] = v;
            }
        }
        double* X = malloc(sizeof(double) * 3 * NT); double* rep = malloc(sizeof(double) * NT); uint8_t* val = malloc(NT);
        CHECK(vo_triangulate_points(ctx, matlab, p0, p1, NT, c0m, c1m, X, rep, val));
        double* Xo = malloc(sizeof(double) * 3 * NT); double* repo = malloc(sizeof(double) * NT); uint8_t* valo = malloc(NT);
        vo_or_triangulate_cam(r0, r1, c0, c1, NT, Xo, repo, valo);
        double worst = 0; int flips = 0, nvalid = 0;
        for (int i = 0; i < NT; ++i) {
            flips += val[i] != valo[i]; nvalid += val[i];
            for (int c = 0; c < 3; ++c) worst = fmax(worst, fabs(X[c * NT + i] - Xo[3 * i + c]) / fmax(1.0, fabs(Xo[3 * i + c])));
            worst = fmax(worst, fabs(rep[i] - repo[i]) / fmax(1.0, repo[i]));
        }
        const int ok = flips == 0 && worst < 1e-12 && nvalid > NT / 2;
        printf("triangulate_points: %d of %d valid, %d validity flips, worst relative difference to the oracle %.1e: %s\n", nvalid, NT, flips, worst,
               ok ? "ok" : "FAILED");
        failures += !ok;
    }
    vo_ctx_destroy(ctx);
    printf("%s\n", failures ? "C HOST: FAILED" : "C HOST: all checks passed");
    return failures ? 1 : 0;
}
