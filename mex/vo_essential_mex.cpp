// [E, inlierIdx, status] = vo_essential_mex(points1, points2, K, opts)
// Gateway behind matlab/estimateEssentialMatrix.m (the toolbox call of modules/bootstrap.m:76-79).
// points M x 2 double, K = cameraParams.IntrinsicMatrix as stored; opts: MaxNumTrials, Confidence (percent),
// MaxDistance (pixels^2, Sampson), Seed, Solver (5: Nister's five-point solver, the default -- the kind the toolbox function
// wraps; 8: the normalised eight-point algorithm).  E comes back as the toolbox returns it (x2' * E * x1 = 0, row vectors).
#include "vo_mex_common.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 4) mexErrMsgIdAndTxt("vo:args", "usage: vo_essential_mex(points1, points2, K, opts)");
    if (!mxIsDouble(prhs[0]) || !mxIsDouble(prhs[1]) || !mxIsDouble(prhs[2])) mexErrMsgIdAndTxt("vo:args", "inputs must be double");
    const int M = (int)mxGetM(prhs[0]);
    if (mxGetM(prhs[1]) != (size_t)M || (M > 0 && (mxGetN(prhs[0]) != 2 || mxGetN(prhs[1]) != 2)) || mxGetNumberOfElements(prhs[2]) != 9)
        mexErrMsgIdAndTxt("vo:args", "shape mismatch");
    vo_ransac_opts r;
    r.num_hyp = (int)vo_opt(prhs[3], "MaxNumTrials", 500);
    r.confidence = vo_opt(prhs[3], "Confidence", 99) / 100.0;
    r.max_reproj_err = vo_opt(prhs[3], "MaxDistance", 0.1);
    r.seed = (uint64_t)vo_opt(prhs[3], "Seed", 1023);
    r.mode = 1;
    const int solver = (int)vo_opt(prhs[3], "Solver", 5);
    if (solver != 5 && solver != 8) mexErrMsgIdAndTxt("vo:args", "Solver must be 5 or 8");
    r.min_inliers = solver;
    plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
    mxArray* inl = mxCreateLogicalMatrix(M, 1);
    vo_ransac_info info;
    if (solver == 5)
        vo_mex_check(vo_essential_ransac_five(vo_mex_ctx(), VO_MATLAB, mxGetPr(prhs[0]), mxGetPr(prhs[1]), M, mxGetPr(prhs[2]), &r, nullptr,
                                              mxGetPr(plhs[0]), (uint8_t*)mxGetData(inl), &info, nullptr, nullptr, nullptr),
                     "vo_essential_ransac_five");
    else
        vo_mex_check(vo_essential_ransac(vo_mex_ctx(), VO_MATLAB, mxGetPr(prhs[0]), mxGetPr(prhs[1]), M, mxGetPr(prhs[2]), &r, nullptr,
                                         nullptr, mxGetPr(plhs[0]), (uint8_t*)mxGetData(inl), &info, nullptr, nullptr, nullptr),
                     "vo_essential_ransac");
    if (nlhs > 1) plhs[1] = inl; else mxDestroyArray(inl);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar((double)info.status);
}
