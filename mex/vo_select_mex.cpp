// loc = vo_select_mex(img, curr_kps, opts)   -> M x 2 single [x y]
// Gateway behind matlab/selectCandidateKeypoints.m (detectMinEigenFeatures + pdist2 + selectUniform,
// modules/selectCandidateKeypoints.m:20-35).  opts: MinDistance, MinQuality, FilterSize,
// CandidatesToKeep, Scorer (0 min-eigen, 1 Harris), Kappa.
#include "vo_mex_common.h"
#include <string.h>

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 3) mexErrMsgIdAndTxt("vo:args", "usage: vo_select_mex(img, curr_kps, opts)");
    const mxArray *im = prhs[0], *kp = prhs[1], *o = prhs[2];
    if (!mxIsUint8(im) || mxGetNumberOfDimensions(im) != 2) mexErrMsgIdAndTxt("vo:args", "img must be 2-D uint8");
    const int H = (int)mxGetM(im), W = (int)mxGetN(im), Ne = (int)mxGetM(kp);
    if (Ne > 0 && (mxGetN(kp) != 2 || !(mxIsDouble(kp) || mxIsSingle(kp)))) mexErrMsgIdAndTxt("vo:args", "curr_kps must be N x 2");
    vo_sel_opts s;
    s.min_distance = vo_opt(o, "MinDistance", 20);
    s.min_quality = vo_opt(o, "MinQuality", 0.01);
    s.filter_size = (int)vo_opt(o, "FilterSize", 5);
    s.keep = (int)vo_opt(o, "CandidatesToKeep", 200);
    s.scorer = (int)vo_opt(o, "Scorer", VO_SCORER_MINEIGEN);
    s.kappa = vo_opt(o, "Kappa", 0.08);
    float* tmp = (float*)mxMalloc(sizeof(float) * 2 * (size_t)(s.keep > 0 ? s.keep : 1));
    int M = 0;
    int rc = vo_select_candidates(vo_mex_ctx(), (const uint8_t*)mxGetData(im), H, W, VO_MATLAB, Ne ? mxGetData(kp) : nullptr, Ne,
                                  mxIsDouble(kp) ? 1 : 0, &s, tmp, nullptr, nullptr, &M);
    if (rc != VO_OK) { mxFree(tmp); vo_mex_check(rc, "vo_select_candidates"); }
    plhs[0] = mxCreateNumericMatrix(M, 2, mxSINGLE_CLASS, mxREAL); // M x 2 column-major: all x, then all y
    if (M > 0) memcpy(mxGetData(plhs[0]), tmp, sizeof(float) * 2 * (size_t)M);
    mxFree(tmp);
}
