// [X, reproErr, angleDeg, isValid] = vo_triang_mex(first_kps, curr_kps, first_poses, curr_pose, K)
// Gateway for the batched candidate loop body (modules/VisualOdometry.m:93-113): first_kps / curr_kps
// C x 2 double, first_poses 4 x 3 x C double (cat(3, candidate_first_poses{:})), curr_pose 4 x 3,
// K = cameraParams.IntrinsicMatrix (K^T, passed as stored).
#include "vo_mex_common.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 5) mexErrMsgIdAndTxt("vo:args", "usage: vo_triang_mex(first_kps, curr_kps, first_poses, curr_pose, K)");
    for (int i = 0; i < 5; ++i)
        if (!mxIsDouble(prhs[i])) mexErrMsgIdAndTxt("vo:args", "all inputs must be double (cast single state first)");
    const int C = (int)mxGetM(prhs[0]);
    const mwSize* d = mxGetDimensions(prhs[2]);
    const int P = C == 0 ? 0 : (mxGetNumberOfDimensions(prhs[2]) == 3 ? (int)d[2] : 1);
    if (C > 0 && (mxGetM(prhs[1]) != (size_t)C || d[0] != 4 || d[1] != 3 || P != C || mxGetNumberOfElements(prhs[3]) != 12 ||
                  mxGetNumberOfElements(prhs[4]) != 9))
        mexErrMsgIdAndTxt("vo:args", "shape mismatch");
    plhs[0] = mxCreateDoubleMatrix(C, 3, mxREAL);
    mxArray* rep = mxCreateDoubleMatrix(C, 1, mxREAL);
    mxArray* ang = mxCreateDoubleMatrix(C, 1, mxREAL);
    mxArray* val = mxCreateLogicalMatrix(C, 1);
    if (C > 0)
        vo_mex_check(vo_triangulate_candidates(vo_mex_ctx(), VO_MATLAB, mxGetPr(prhs[0]), mxGetPr(prhs[1]), mxGetPr(prhs[2]), P,
                                               nullptr, C, mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(plhs[0]), mxGetPr(rep),
                                               mxGetPr(ang), (uint8_t*)mxGetData(val)), "vo_triangulate_candidates");
    if (nlhs > 1) plhs[1] = rep; else mxDestroyArray(rep);
    if (nlhs > 2) plhs[2] = ang; else mxDestroyArray(ang);
    if (nlhs > 3) plhs[3] = val; else mxDestroyArray(val);
}
