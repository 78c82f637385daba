function [E, inliersIndex, status] = estimateEssentialMatrix(matchedPoints1, matchedPoints2, cameraParams, nv)
% Same-named replacement for the Computer Vision Toolbox call at modules/bootstrap.m:76-79: seeded MSAC with the
% Sampson distance on the GPU (vo_essential_mex) around Nister's five-point solver (Solver = 5, the default: the kind
% of solver the toolbox function wraps) or the normalised eight-point algorithm (Solver = 8).
% status: 0 ok, 1 fewer points than a sample takes, 2 too few inliers.
arguments
    matchedPoints1
    matchedPoints2
    cameraParams
    nv.MaxNumTrials double = 500
    nv.Confidence double = 99
    nv.MaxDistance double = 0.1
    nv.Seed double = 1023
    nv.Solver double = 5
end
[E, inliersIndex, status] = vo_essential_mex(double(matchedPoints1), double(matchedPoints2), ...
    double(cameraParams.IntrinsicMatrix), nv);
end
