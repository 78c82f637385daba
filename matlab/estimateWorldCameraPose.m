function [R_WC, T_WC, inlierIdx, status] = estimateWorldCameraPose(imagePoints, worldPoints, cameraParams, nv)
% Same-named replacement for the Computer Vision Toolbox call at modules/VisualOdometry.m:285-289:
% seeded P3P + MSAC on the GPU (vo_pose_mex).  status: 0 ok, 1 fewer than 4 points, 2 too few inliers.
arguments
    imagePoints
    worldPoints
    cameraParams
    nv.MaxNumTrials double = 1000
    nv.Confidence double = 99
    nv.MaxReprojectionError double = 1
    nv.Seed double = 1023
end
[R_WC, T_WC, inlierIdx, status] = vo_pose_mex(double(imagePoints), double(worldPoints), ...
    double(cameraParams.IntrinsicMatrix), nv);
end
