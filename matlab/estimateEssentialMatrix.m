function [E, inliersIndex, status] = estimateEssentialMatrix(matchedPoints1, matchedPoints2, cameraParams, nv)
% Same-named replacement for the Computer Vision Toolbox call at modules/bootstrap.m:76-79: seeded MSAC around
% the normalised eight-point algorithm with the Sampson distance on the GPU (vo_essential_mex).
% status: 0 ok, 1 fewer than 8 points, 2 too few inliers.
arguments
    matchedPoints1
    matchedPoints2
    cameraParams
    nv.MaxNumTrials double = 500
    nv.Confidence double = 99
    nv.MaxDistance double = 0.1
    nv.Seed double = 1023
end
[E, inliersIndex, status] = vo_essential_mex(double(matchedPoints1), double(matchedPoints2), ...
    double(cameraParams.IntrinsicMatrix), nv);
end
