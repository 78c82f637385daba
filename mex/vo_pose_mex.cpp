// [R_WC, T_WC, inlierIdx, status] = vo_pose_mex(imagePoints, worldPoints, K, opts)
// Gateway behind matlab/estimateWorldCameraPose.m (the toolbox call of modules/VisualOdometry.m:285-289).
// imagePoints M x 2 double, worldPoints M x 3 double, K = cameraParams.IntrinsicMatrix as stored,
// opts: MaxNumTrials, Confidence (percent), MaxReprojectionError, Seed.
#include "vo_mex_common.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 4) mexErrMsgIdAndTxt("vo:args", "usage: vo_pose_mex(imagePoints, worldPoints, K, opts)");
    if (!mxIsDouble(prhs[0]) || !mxIsDouble(prhs[1]) || !mxIsDouble(prhs[2])) mexErrMsgIdAndTxt("vo:args", "inputs must be double");
    const int M = (int)mxGetM(prhs[0]);
    if (mxGetM(prhs[1]) != (size_t)M || (M > 0 && (mxGetN(prhs[0]) != 2 || mxGetN(prhs[1]) != 3)) || mxGetNumberOfElements(prhs[2]) != 9)
        mexErrMsgIdAndTxt("vo:args", "shape mismatch");
    vo_ransac_opts r;
    r.num_hyp = (int)vo_opt(prhs[3], "MaxNumTrials", 1000);
    r.confidence = vo_opt(prhs[3], "Confidence", 99) / 100.0;
    r.max_reproj_err = vo_opt(prhs[3], "MaxReprojectionError", 1);
    r.seed = (uint64_t)vo_opt(prhs[3], "Seed", 1023); // module_tests/VisualOdometry_test.m:5 rng(1023)
    r.mode = 1;        // MSAC, like estimateWorldCameraPose
    r.min_inliers = 4;
    plhs[0] = mxCreateDoubleMatrix(3, 3, mxREAL);
    mxArray* t = mxCreateDoubleMatrix(1, 3, mxREAL);
    mxArray* inl = mxCreateLogicalMatrix(M, 1);
    vo_ransac_info info;
    vo_mex_check(vo_p3p_ransac(vo_mex_ctx(), VO_MATLAB, mxGetPr(prhs[0]), mxGetPr(prhs[1]), M, mxGetPr(prhs[2]), &r, nullptr, nullptr,
                               mxGetPr(plhs[0]), mxGetPr(t), (uint8_t*)mxGetData(inl), &info, nullptr, nullptr, nullptr),
                 "vo_p3p_ransac");
    if (nlhs > 1) plhs[1] = t; else mxDestroyArray(t);
    if (nlhs > 2) plhs[2] = inl; else mxDestroyArray(inl);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)info.status);
}
