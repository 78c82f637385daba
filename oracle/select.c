/* oracle/select.c -- CPU restatement of candidate keypoint selection (TEST INFRASTRUCTURE).
 *
 * Reference: modules/selectCandidateKeypoints.m:20-35 (detectMinEigenFeatures -> pdist2 ->
 * strict-< distance filter -> selectUniform -> .Location); old/harris.m:3-23 and
 * old/selectKeypoints.m:5-14 for the Harris scorer and the greedy box-suppression NMS.
 * detectMinEigenFeatures / selectUniform are closed source: normative definitions N6-N8 of
 * vo_oracle.h.  Integer structure tensor + a fixed sequence of IEEE double operations make every
 * decision reproducible bit-for-bit.
 */
#include "vo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* Threads for the loops whose iterations are independent (response rows, distance filter; geom.c: candidates).
 * 1 = the serial loops; the results do not depend on it.  Set by bench.py's cpu_baseline leg only. */
int vo_or_threads = 1;
void vo_or_set_threads(int n) { vo_or_threads = n > 0 ? n : 1; }

/* N6 integer Gaussian weights; w has filter_size entries */
void vo_or_gauss_weights(int fs, int32_t* w) {
    int r = fs / 2;
    double s = (double)fs / 3.0, sum = 0, g[64];
    for (int k = -r; k <= r; ++k) { g[k + r] = exp(-(double)(k * k) / (2.0 * s * s)); sum += g[k + r]; }
    for (int k = 0; k < fs; ++k) w[k] = (int32_t)lround(4096.0 * g[k] / sum);
}

static void response_mineigen(const uint8_t* img, int nu, int nv, int fs, double* resp) {
    const int r = fs / 2;
    int32_t w[64];
    vo_or_gauss_weights(fs, w);
    long n = (long)nu * nv;
    int32_t* a = (int32_t*)malloc(sizeof(int32_t) * n * 3);
    int32_t *b = a + n, *c = b + n;
#pragma omp parallel for num_threads(vo_or_threads) schedule(static)
    for (int v = 0; v < nv; ++v)
        for (int u = 0; u < nu; ++u) {
            int du = (int)img[(long)v * nu + clampi(u + 1, 0, nu - 1)] - (int)img[(long)v * nu + clampi(u - 1, 0, nu - 1)];
            int dv = (int)img[(long)clampi(v + 1, 0, nv - 1) * nu + u] - (int)img[(long)clampi(v - 1, 0, nv - 1) * nu + u];
            a[(long)v * nu + u] = du * du; b[(long)v * nu + u] = du * dv; c[(long)v * nu + u] = dv * dv;
        }
    int64_t* h = (int64_t*)malloc(sizeof(int64_t) * n * 3);
    int64_t *hb = h + n, *hc = hb + n;
#pragma omp parallel for num_threads(vo_or_threads) schedule(static)
    for (int v = 0; v < nv; ++v)
        for (int u = 0; u < nu; ++u) {
            int64_t sa = 0, sb = 0, sc = 0;
            for (int k = -r; k <= r; ++k) {
                long idx = (long)v * nu + clampi(u + k, 0, nu - 1);
                sa += (int64_t)w[k + r] * a[idx]; sb += (int64_t)w[k + r] * b[idx]; sc += (int64_t)w[k + r] * c[idx];
            }
            h[(long)v * nu + u] = sa; hb[(long)v * nu + u] = sb; hc[(long)v * nu + u] = sc;
        }
#pragma omp parallel for num_threads(vo_or_threads) schedule(static)
    for (int v = 0; v < nv; ++v)
        for (int u = 0; u < nu; ++u) {
            int64_t sa = 0, sb = 0, sc = 0;
            for (int k = -r; k <= r; ++k) {
                long idx = (long)clampi(v + k, 0, nv - 1) * nu + u;
                sa += (int64_t)w[k + r] * h[idx]; sb += (int64_t)w[k + r] * hb[idx]; sc += (int64_t)w[k + r] * hc[idx];
            }
            double A = (double)sa, B = (double)sb, C = (double)sc;
            double dif = A - C;
            double rad = dif * dif + 4.0 * (B * B);
            double lam = 0.5 * ((A + C) - sqrt(rad));
            if (lam < 0.0) lam = 0.0;
            resp[(long)v * nu + u] = lam * (1.0 / 67108864.0); /* 2^-26 */
        }
    free(a); free(h);
}

/* old/harris.m:3-23.  The image there is indexed (row, col); conv2(col_kernel, row_kernel, img).
 * In internal coordinates the first conv2 kernel runs along the FAST axis when the image is
 * MATLAB column-major (u = row).  Ix*Iy and the squares are invariant to the sign flip of true
 * convolution, and the response is symmetric under swapping the roles of Ix and Iy, so the
 * result is the same function of the neighbourhood for either memory layout. */
static void response_harris(const uint8_t* img, int nu, int nv, int patch, double kappa, double* resp) {
    const int pr = patch / 2;
    long n = (long)nu * nv;
    memset(resp, 0, sizeof(double) * n);
    if (nu < 3 || nv < 3) return;
    int su = nu - 2, sv = nv - 2; /* 'valid' Sobel output, harris.m:6-7 */
    double* xx = (double*)malloc(sizeof(double) * (long)su * sv * 3);
    double *yy = xx + (long)su * sv, *xy = yy + (long)su * sv;
    for (int v = 0; v < sv; ++v)
        for (int u = 0; u < su; ++u) {
#define P(a, b) ((double)img[(long)(v + (b)) * nu + (u + (a))])
            double ix = (P(0, 0) + 2.0 * P(0, 1) + P(0, 2)) - (P(2, 0) + 2.0 * P(2, 1) + P(2, 2));
            double iy = (P(0, 0) + 2.0 * P(1, 0) + P(2, 0)) - (P(0, 2) + 2.0 * P(1, 2) + P(2, 2));
#undef P
            xx[(long)v * su + u] = ix * ix; yy[(long)v * su + u] = iy * iy; xy[(long)v * su + u] = ix * iy; /* :8-10 */
        }
    int bu = su - patch + 1, bv = sv - patch + 1; /* :14-16 */
    for (int v = 0; v < bv; ++v)
        for (int u = 0; u < bu; ++u) {
            double sxx = 0, syy = 0, sxy = 0;
            for (int j = 0; j < patch; ++j)
                for (int i = 0; i < patch; ++i) {
                    long idx = (long)(v + j) * su + (u + i);
                    sxx += xx[idx]; syy += yy[idx]; sxy += xy[idx];
                }
            double tr = sxx + syy;
            double sc = (sxx * syy - sxy * sxy) - kappa * (tr * tr); /* :18-19 */
            if (sc < 0) sc = 0;                                    /* :21 */
            resp[(long)(v + 1 + pr) * nu + (u + 1 + pr)] = sc;     /* :23 */
        }
    free(xx);
}

void vo_or_corner_response(const uint8_t* img, int nu, int nv, int scorer, int fs, double kappa, double* resp) {
    if (scorer == VO_SCORER_HARRIS) response_harris(img, nu, nv, fs, kappa, resp);
    else response_mineigen(img, nu, nv, fs, resp);
}

/* N7 */
int vo_or_detect_corners(const double* resp, int nu, int nv, double min_quality, int max_out,
                         int32_t* lin_idx, float* loc_u, float* loc_v, double* metric) {
    long n = (long)nu * nv;
    double mx = 0;
    for (long i = 0; i < n; ++i) if (resp[i] > mx) mx = resp[i];
    const double thr = min_quality * mx;
    int m = 0;
    for (int v = 1; v < nv - 1; ++v)
        for (int u = 1; u < nu - 1; ++u) {
            const long i = (long)v * nu + u;
            const double r = resp[i];
            if (!(r > thr)) continue;
            int ismax = 1;
            for (int dv = -1; dv <= 1 && ismax; ++dv)
                for (int du = -1; du <= 1; ++du) {
                    if (!du && !dv) continue;
                    long j = (long)(v + dv) * nu + (u + du);
                    double q = resp[j];
                    if (q > r || (q == r && j < i)) { ismax = 0; break; }
                }
            if (!ismax) continue;
            double rum = resp[i - 1], rup = resp[i + 1], rvm = resp[i - nu], rvp = resp[i + nu];
            double cuu = (rup - 2.0 * r) + rum, cvv = (rvp - 2.0 * r) + rvm;
            double ou = 0, ov = 0;
            if (cuu < 0) { ou = -0.5 * (rup - rum) / cuu; if (ou > 0.5) ou = 0.5; if (ou < -0.5) ou = -0.5; }
            if (cvv < 0) { ov = -0.5 * (rvp - rvm) / cvv; if (ov > 0.5) ov = 0.5; if (ov < -0.5) ov = -0.5; }
            if (m < max_out) {
                lin_idx[m] = (int32_t)i;
                loc_u[m] = (float)((double)u + ou);
                loc_v[m] = (float)((double)v + ov);
                metric[m] = r;
            }
            ++m;
        }
    return m;
}

/* N8, modules/selectCandidateKeypoints.m:23-29 */
void vo_or_distance_filter(const float* cu, const float* cv, int m, const float* eu, const float* ev,
                           int ne, float mind, uint8_t* keep) {
#pragma omp parallel for num_threads(vo_or_threads) schedule(dynamic, 64)
    for (int i = 0; i < m; ++i) {
        int k = 1;
        for (int j = 0; j < ne; ++j) {
            float dx = cu[i] - eu[j], dy = cv[i] - ev[j];
            float d2 = dx * dx + dy * dy;
            if (sqrtf(d2) < mind) { k = 0; break; }
        }
        keep[i] = (uint8_t)k;
    }
}

/* N8 selectUniform.  Grid of (cells_u x cells_v) per round; the caller-visible image is H x W with
 * aspect = W/H; `u_is_rows` says whether the internal u axis is the row (y) axis (MATLAB layout). */
int vo_or_select_uniform(const float* cu, const float* cv, const double* metric, const uint8_t* keep,
                         int m, int n_keep, int nu, int nv, int u_is_rows, uint8_t* sel) {
    memset(sel, 0, (size_t)m);
    int remaining = 0;
    for (int i = 0; i < m; ++i) remaining += keep[i] ? 1 : 0;
    int selected = 0;
    const long H = u_is_rows ? nu : nv, W = u_is_rows ? nv : nu;
    while (selected < n_keep && remaining > 0) {
        long want = n_keep - selected;
        /* h = floor(sqrt(want / (W/H))), w = floor(h * W/H) in exact integer arithmetic */
        long x = (want * H) / W, h = 0;
        while ((h + 1) * (h + 1) <= x) ++h;
        if (h < 1) h = 1;
        if (h > want) h = want;
        long w = (h * W) / H; if (w < 1) w = 1; if (h * w > want) w = want / h;
        int cells_u = (int)(u_is_rows ? h : w), cells_v = (int)(u_is_rows ? w : h);
        int nc = cells_u * cells_v;
        int* best = (int*)malloc(sizeof(int) * nc);
        for (int k = 0; k < nc; ++k) best[k] = -1;
        for (int i = 0; i < m; ++i) {
            if (!keep[i] || sel[i]) continue;
            int bu = (int)floor(((double)cu[i] * (double)cells_u) / (double)nu);
            int bv = (int)floor(((double)cv[i] * (double)cells_v) / (double)nv);
            bu = clampi(bu, 0, cells_u - 1); bv = clampi(bv, 0, cells_v - 1);
            int cell = bv * cells_u + bu;
            if (best[cell] < 0 || metric[i] > metric[best[cell]]) best[cell] = i;
        }
        /* if the round would overshoot (cells > want cannot happen since h*w <= want) */
        for (int k = 0; k < nc; ++k)
            if (best[k] >= 0) { sel[best[k]] = 1; ++selected; --remaining; }
        free(best);
    }
    return selected;
}

int vo_or_select_candidates(const uint8_t* img, int nu, int nv, const float* eu, const float* ev,
                            int ne, float mind, double minq, int fs, int keepn, int scorer,
                            double kappa, int col_major, float* out_u, float* out_v,
                            int32_t* out_lin, double* out_metric) {
    long n = (long)nu * nv;
    double* resp = (double*)malloc(sizeof(double) * n);
    vo_or_corner_response(img, nu, nv, scorer, fs, kappa, resp);
    int cap = (int)(n / 4 + 16);
    int32_t* lin = (int32_t*)malloc(sizeof(int32_t) * cap);
    float* lu = (float*)malloc(sizeof(float) * cap * 2);
    float* lv = lu + cap;
    double* met = (double*)malloc(sizeof(double) * cap);
    int m = vo_or_detect_corners(resp, nu, nv, minq, cap, lin, lu, lv, met);
    if (m > cap) m = cap;
    uint8_t* keep = (uint8_t*)malloc((size_t)m * 2 + 2);
    uint8_t* sel = keep + m + 1;
    vo_or_distance_filter(lu, lv, m, eu, ev, ne, mind, keep);
    vo_or_select_uniform(lu, lv, met, keep, m, keepn, nu, nv, col_major, sel);
    int k = 0;
    for (int i = 0; i < m; ++i)
        if (sel[i]) { out_u[k] = lu[i]; out_v[k] = lv[i]; out_lin[k] = lin[i]; if (out_metric) out_metric[k] = met[i]; ++k; }
    free(resp); free(lin); free(lu); free(met); free(keep);
    return k;
}

/* old/selectKeypoints.m:5-14.  scores is nu x nv internal (u = MATLAB row when column-major, so
 * the internal linear index IS MATLAB's column-major linear index and max() picks the first). */
void vo_or_select_keypoints_greedy(const double* scores, int nu, int nv, int num, int r,
                                   int32_t* out_u, int32_t* out_v) {
    int pu = nu + 2 * r, pv = nv + 2 * r;
    double* t = (double*)calloc((size_t)pu * pv, sizeof(double)); /* padarray, :6 */
    for (int v = 0; v < nv; ++v)
        for (int u = 0; u < nu; ++u) t[(long)(v + r) * pu + (u + r)] = scores[(long)v * nu + u];
    for (int i = 0; i < num; ++i) {
        long best = 0; double bv = t[0];
        for (long k = 1; k < (long)pu * pv; ++k) if (t[k] > bv) { bv = t[k]; best = k; } /* :8 */
        int ku = (int)(best % pu), kv = (int)(best / pu);
        out_u[i] = ku - r; out_v[i] = kv - r; /* :11, 0-based */
        for (int dv = -r; dv <= r; ++dv)
            for (int du = -r; du <= r; ++du) {
                int a = ku + du, b = kv + dv;
                if (a >= 0 && a < pu && b >= 0 && b < pv) t[(long)b * pu + a] = 0; /* :12-13 */
            }
    }
    free(t);
}
