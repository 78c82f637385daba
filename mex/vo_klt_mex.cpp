// [points, validity, scores] = vo_klt_mex(img1, img2, pts, opts)
// Gateway behind matlab/KLTTracker.m (replaces vision.PointTracker in modules/KLTTracker.m:37-39).
// opts: struct with NumPyramidLevels, MaxBidirectionalError, BlockSize [h w], MaxIterations.
#include "vo_mex_common.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 4) mexErrMsgIdAndTxt("vo:args", "usage: vo_klt_mex(img1, img2, pts, opts)");
    const mxArray *i1 = prhs[0], *i2 = prhs[1], *p = prhs[2], *o = prhs[3];
    if (!mxIsUint8(i1) || !mxIsUint8(i2) || mxGetNumberOfDimensions(i1) != 2)
        mexErrMsgIdAndTxt("vo:args", "images must be 2-D uint8 (grayscale)");
    const int H = (int)mxGetM(i1), W = (int)mxGetN(i1);
    if ((int)mxGetM(i2) != H || (int)mxGetN(i2) != W) mexErrMsgIdAndTxt("vo:args", "images differ in size");
    const int N = (int)mxGetM(p);
    if (N == 0) { // modules/KLTTracker.m:33-36
        plhs[0] = mxCreateDoubleMatrix(0, 2, mxREAL);
        if (nlhs > 1) plhs[1] = mxCreateLogicalMatrix(0, 0);
        if (nlhs > 2) plhs[2] = mxCreateDoubleMatrix(0, 1, mxREAL);
        return;
    }
    if (mxGetN(p) != 2 || !(mxIsDouble(p) || mxIsSingle(p))) mexErrMsgIdAndTxt("vo:args", "pts must be N x 2 single/double");
    const bool dbl = mxIsDouble(p);
    const mxClassID cls = dbl ? mxDOUBLE_CLASS : mxSINGLE_CLASS; // outputs follow the class of the points
    vo_klt_opts k;
    k.levels = (int)vo_opt(o, "NumPyramidLevels", 4);
    k.max_bidir_err = vo_opt(o, "MaxBidirectionalError", 2);
    const mxArray* bs = mxGetField(o, 0, "BlockSize");
    k.block_h = bs ? (int)mxGetPr(bs)[0] : 41;
    k.block_w = bs ? (int)mxGetPr(bs)[1] : 41;
    k.max_iters = (int)vo_opt(o, "MaxIterations", 50);
    k.eps = 0.01;
    plhs[0] = mxCreateNumericMatrix(N, 2, cls, mxREAL);
    mxArray* valid = mxCreateLogicalMatrix(N, 1);
    mxArray* score = nlhs > 2 ? mxCreateNumericMatrix(N, 1, cls, mxREAL) : nullptr; // scores are only computed when asked for
    vo_mex_check(vo_klt_track(vo_mex_ctx(), (const uint8_t*)mxGetData(i1), (const uint8_t*)mxGetData(i2), H, W, VO_MATLAB,
                              mxGetData(p), N, dbl ? 1 : 0, &k, mxGetData(plhs[0]), (uint8_t*)mxGetData(valid),
                              score ? mxGetData(score) : nullptr), "vo_klt_track");
    if (nlhs > 1) plhs[1] = valid; else mxDestroyArray(valid);
    if (nlhs > 2) plhs[2] = score;
}
