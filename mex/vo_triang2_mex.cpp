// [worldPoints, reprojectionErrors, validIndex] = vo_triang2_mex(points1, points2, camMatrix1, camMatrix2)
// Gateway for the toolbox call `triangulate` as the reference makes it (modules/VisualOdometry.m:97-101,
// modules/candidateTriangulation.m:47-51, processFrame.m:55-59): points N x 2 double, camera matrices 4 x 3 double
// (what `cameraMatrix` returns).  matlab/triangulate.m casts and forwards.
#include "vo_mex_common.h"

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 4) mexErrMsgIdAndTxt("vo:args", "usage: vo_triang2_mex(points1, points2, camMatrix1, camMatrix2)");
    for (int i = 0; i < 4; ++i)
        if (!mxIsDouble(prhs[i])) mexErrMsgIdAndTxt("vo:args", "all inputs must be double (cast single points first)");
    const int n = (int)mxGetM(prhs[0]);
    if (mxGetN(prhs[0]) != 2 || mxGetM(prhs[1]) != (size_t)n || mxGetN(prhs[1]) != 2 || mxGetM(prhs[2]) != 4 || mxGetN(prhs[2]) != 3 ||
        mxGetM(prhs[3]) != 4 || mxGetN(prhs[3]) != 3)
        mexErrMsgIdAndTxt("vo:args", "points must be N x 2 (the same N), camera matrices 4 x 3");
    plhs[0] = mxCreateDoubleMatrix(n, 3, mxREAL);
    mxArray* rep = mxCreateDoubleMatrix(n, 1, mxREAL);
    mxArray* val = mxCreateLogicalMatrix(n, 1);
    if (n > 0)
        vo_mex_check(vo_triangulate_points(vo_mex_ctx(), VO_MATLAB, mxGetPr(prhs[0]), mxGetPr(prhs[1]), n, mxGetPr(prhs[2]),
                                           mxGetPr(prhs[3]), mxGetPr(plhs[0]), mxGetPr(rep), (uint8_t*)mxGetData(val)),
                     "vo_triangulate_points");
    if (nlhs > 1) plhs[1] = rep; else mxDestroyArray(rep);
    if (nlhs > 2) plhs[2] = val; else mxDestroyArray(val);
}
