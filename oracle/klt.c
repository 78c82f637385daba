/* oracle/klt.c -- CPU restatement of KLTTracker.track (TEST INFRASTRUCTURE, see vo_oracle.h).
 *
 * Reference: modules/KLTTracker.m:22-41 wraps vision.PointTracker (closed source), options at
 * :11-14, instantiated at modules/VisualOdometry.m:33-37.  The toolbox internals are not
 * observable; the algorithm below is normative definitions N1-N5 of vo_oracle.h (Bouguet's
 * pyramidal Lucas-Kanade with forward-backward check, as the toolbox documents it).
 */
#include "vo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

long vo_or_pyramid_layout(int nu, int nv, int levels, int* nus, int* nvs, long* offs) {
    long tot = 0;
    for (int l = 0; l < levels; ++l) {
        nus[l] = nu; nvs[l] = nv; offs[l] = tot;
        tot += (long)nu * nv;
        nu = (nu + 1) / 2; nv = (nv + 1) / 2;
    }
    return tot;
}

/* N1 */
static void pyr_down(const double* src, int nu, int nv, double* dst) {
    static const double w[5] = {1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16};
    int mu = (nu + 1) / 2, mv = (nv + 1) / 2;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)mu * nv);
    for (int v = 0; v < nv; ++v)
        for (int i = 0; i < mu; ++i) {
            double s = 0;
            for (int a = 0; a < 5; ++a) s += w[a] * src[(long)v * nu + clampi(2 * i + a - 2, 0, nu - 1)];
            tmp[(long)v * mu + i] = s;
        }
    for (int j = 0; j < mv; ++j)
        for (int i = 0; i < mu; ++i) {
            double s = 0;
            for (int b = 0; b < 5; ++b) s += w[b] * tmp[(long)clampi(2 * j + b - 2, 0, nv - 1) * mu + i];
            dst[(long)j * mu + i] = s;
        }
    free(tmp);
}

void vo_or_build_pyramid(const uint8_t* img, int nu, int nv, int levels, double* pyr) {
    int nus[VO_MAX_LEVELS], nvs[VO_MAX_LEVELS]; long offs[VO_MAX_LEVELS];
    vo_or_pyramid_layout(nu, nv, levels, nus, nvs, offs);
    for (long i = 0; i < (long)nu * nv; ++i) pyr[i] = (double)img[i];
    for (int l = 1; l < levels; ++l) pyr_down(pyr + offs[l - 1], nus[l - 1], nvs[l - 1], pyr + offs[l]);
}

/* N2 (Sobel/8 of a whole plane; the tracker applies the same operator to the sampled template) */
void vo_or_gradients(const double* I, int nu, int nv, double* gu, double* gv) {
    for (int v = 0; v < nv; ++v) {
        int vm = clampi(v - 1, 0, nv - 1), vp = clampi(v + 1, 0, nv - 1);
        for (int u = 0; u < nu; ++u) {
            int um = clampi(u - 1, 0, nu - 1), up = clampi(u + 1, 0, nu - 1);
#define P(a, b) I[(long)(b) * nu + (a)]
            double du = (P(up, vm) - P(um, vm)) + 2.0 * (P(up, v) - P(um, v)) + (P(up, vp) - P(um, vp));
            double dv = (P(um, vp) - P(um, vm)) + 2.0 * (P(u, vp) - P(u, vm)) + (P(up, vp) - P(up, vm));
#undef P
            gu[(long)v * nu + u] = du * 0.125;
            gv[(long)v * nu + u] = dv * 0.125;
        }
    }
}

/* N3 */
static inline double bilin(const double* I, int nu, int nv, double u, double v) {
    double fl_u = floor(u), fl_v = floor(v);
    double fu = u - fl_u, fv = v - fl_v;
    int iu = (int)fl_u, iv = (int)fl_v;
    int u0 = clampi(iu, 0, nu - 1), u1 = clampi(iu + 1, 0, nu - 1);
    int v0 = clampi(iv, 0, nv - 1), v1 = clampi(iv + 1, 0, nv - 1);
    double w00 = (1 - fu) * (1 - fv), w10 = fu * (1 - fv), w01 = (1 - fu) * fv, w11 = fu * fv;
    return w00 * I[(long)v0 * nu + u0] + w10 * I[(long)v0 * nu + u1] +
           w01 * I[(long)v1 * nu + u0] + w11 * I[(long)v1 * nu + u1];
}

typedef struct {
    int levels, nus[VO_MAX_LEVELS], nvs[VO_MAX_LEVELS];
    long offs[VO_MAX_LEVELS];
    double* I;
} pyr_t;

static void pyr_make(pyr_t* p, const uint8_t* img, int nu, int nv, int levels) {
    p->levels = levels;
    long tot = vo_or_pyramid_layout(nu, nv, levels, p->nus, p->nvs, p->offs);
    p->I = (double*)malloc(sizeof(double) * tot);
    vo_or_build_pyramid(img, nu, nv, levels, p->I);
}
static void pyr_free(pyr_t* p) { free(p->I); }

/* N4: one direction A -> B.  returns flags: bit0 lost, bit1 flat at level 0 */
static int lk_one(const pyr_t* A, const pyr_t* B, double pu, double pv, const vo_or_klt_opts* o,
                  double* T, double* Tu, double* Tv, double* E, double* qu, double* qv, double* ssd_out,
                  int* iters) {
    const int L = o->levels, wu = o->wu, wv = o->wv, npx = wu * wv;
    const int ru = (wu - 1) / 2, rv = (wv - 1) / 2;
    const int nu0 = A->nus[0], nv0 = A->nvs[0];
    double gu_ = 0, gv_ = 0;
    int flags = 0;
    for (int l = L - 1; l >= 0; --l) {
        const double s = (double)(1 << l);
        const int nu = A->nus[l], nv = A->nvs[l];
        const double* AI = A->I + A->offs[l];
        const double* BI = B->I + B->offs[l];
        const double plu = pu / s, plv = pv / s;
        double Guu = 0, Guv = 0, Gvv = 0;
        /* N4: template sampled on the window grown by one pixel; gradients = Sobel/8 of the samples */
        const int ew = wu + 2;
        for (int j = 0; j < wv + 2; ++j)
            for (int i = 0; i < ew; ++i)
                E[j * ew + i] = bilin(AI, nu, nv, plu + (double)(i - ru - 1), plv + (double)(j - rv - 1));
        for (int j = 0; j < wv; ++j)
            for (int i = 0; i < wu; ++i) {
                int k = j * wu + i;
#define EE(a, b) E[(j + (b)) * ew + (i + (a))]
                T[k] = EE(1, 1);
                Tu[k] = ((EE(2, 0) - EE(0, 0)) + 2.0 * (EE(2, 1) - EE(0, 1)) + (EE(2, 2) - EE(0, 2))) * 0.125;
                Tv[k] = ((EE(0, 2) - EE(0, 0)) + 2.0 * (EE(1, 2) - EE(1, 0)) + (EE(2, 2) - EE(2, 0))) * 0.125;
#undef EE
                Guu += Tu[k] * Tu[k]; Guv += Tu[k] * Tv[k]; Gvv += Tv[k] * Tv[k];
            }
        const double dif = Guu - Gvv;
        const double mineig = ((Guu + Gvv) - sqrt(dif * dif + 4.0 * (Guv * Guv))) / (2.0 * (double)npx);
        const double det = Guu * Gvv - Guv * Guv;
        double du = 0, dv = 0;
        const int flat = !(mineig >= VO_MIN_EIG);
        if (flat) {
            if (l == 0) flags |= 2;
        } else {
            for (int k = 0; k < o->max_iters; ++k) {
                const double cu = plu + gu_ + du, cv = plv + gv_ + dv;
                if (!(cu * s >= 0.0 && cu * s <= (double)(nu0 - 1) && cv * s >= 0.0 && cv * s <= (double)(nv0 - 1))) {
                    flags |= 1; break;
                }
                double bu = 0, bv = 0;
                for (int j = 0; j < wv; ++j)
                    for (int i = 0; i < wu; ++i) {
                        int kk = j * wu + i;
                        double J = bilin(BI, nu, nv, cu + (double)(i - ru), cv + (double)(j - rv));
                        double e = T[kk] - J;
                        bu += e * Tu[kk]; bv += e * Tv[kk];
                    }
                ++*iters;
                const double ddu = (Gvv * bu - Guv * bv) / det;
                const double ddv = (Guu * bv - Guv * bu) / det;
                du += ddu; dv += ddv;
                if (ddu * ddu + ddv * ddv < o->eps * o->eps) break;
            }
        }
        if (flags & 1) break;
        if (l > 0) { gu_ = 2.0 * (gu_ + du); gv_ = 2.0 * (gv_ + dv); }
        else { gu_ = gu_ + du; gv_ = gv_ + dv; }
    }
    *qu = pu + gu_; *qv = pv + gv_;
    if (!(flags & 1)) {
        if (!(*qu >= 0.0 && *qu <= (double)(nu0 - 1) && *qv >= 0.0 && *qv <= (double)(nv0 - 1))) flags |= 1;
    }
    *ssd_out = 0;
    if (!(flags & 1)) {
        /* N5 score residual at level 0 (T holds the level-0 template) */
        double ssd = 0;
        const double* BI = B->I;
        for (int j = 0; j < wv; ++j)
            for (int i = 0; i < wu; ++i) {
                double J = bilin(BI, nu0, nv0, *qu + (double)(i - ru), *qv + (double)(j - rv));
                double e = T[j * wu + i] - J;
                ssd += e * e;
            }
        *ssd_out = ssd;
    }
    return flags;
}

int vo_or_klt_track(const uint8_t* img1, const uint8_t* img2, int nu, int nv, const double* pu,
                    const double* pv, int n, const vo_or_klt_opts* o, double* qu, double* qv,
                    uint8_t* valid, double* score, double* fb_err, int* iters_out) {
    if (o->levels < 1 || o->levels > VO_MAX_LEVELS || !(o->wu & 1) || !(o->wv & 1)) return -1;
    if (n <= 0) return 0;
    pyr_t A, B;
    pyr_make(&A, img1, nu, nv, o->levels);
    pyr_make(&B, img2, nu, nv, o->levels);
    const int npx = o->wu * o->wv;
    int nth = o->threads > 0 ? o->threads : 1;
    (void)nth;
#pragma omp parallel num_threads(nth)
    {
        double* T = (double*)malloc(sizeof(double) * (3 * npx + (o->wu + 2) * (o->wv + 2)));
        double *Tu = T + npx, *Tv = T + 2 * npx, *E = T + 3 * npx;
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < n; ++i) {
            double fu, fv, ssd, bu = 0, bv = 0, ssd2;
            int it = 0;
            int fl = lk_one(&A, &B, pu[i], pv[i], o, T, Tu, Tv, E, &fu, &fv, &ssd, &it);
            int ok = (fl == 0);
            double fb = NAN;
            if (!(fl & 1) && o->max_bidir >= 0) {
                int fl2 = lk_one(&B, &A, fu, fv, o, T, Tu, Tv, E, &bu, &bv, &ssd2, &it);
                if (fl2 & 1) ok = 0;
                else {
                    double eu = pu[i] - bu, ev = pv[i] - bv;
                    fb = sqrt(eu * eu + ev * ev);
                    if (!(fb <= o->max_bidir)) ok = 0;
                }
            }
            qu[i] = fu; qv[i] = fv;
            valid[i] = (uint8_t)ok;
            double sc = 0;
            if (!(fl & 1)) {
                sc = 1.0 - sqrt(ssd / (double)npx) / 255.0;
                if (sc < 0) sc = 0;
                if (sc > 1) sc = 1;
            }
            score[i] = sc;
            if (fb_err) fb_err[i] = fb;
            if (iters_out) iters_out[i] = it;
        }
        free(T);
    }
    pyr_free(&A); pyr_free(&B);
    return 0;
}
