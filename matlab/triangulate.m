function [worldPoints, reprojectionErrors, validIndex] = triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2)
% Same-named replacement for the Computer Vision Toolbox call at modules/VisualOdometry.m:97-101,
% modules/candidateTriangulation.m:47-51 and processFrame.m:55-59: two-view linear triangulation on the GPU
% (vo_triang2_mex -> vo_triangulate_points).  With this file on the path the UNMODIFIED class method
% VisualOdometry.candidateTriangulation reaches the GPU -- one candidate a call, as its loop is written; the batched
% path is the free function candidateTriangulation (matlab/candidateTriangulation.m).
% Outputs follow the class of the points (single after the first frames, SURVEY 8: dtype drift).
cls = class(matchedPoints1);
[worldPoints, reprojectionErrors, validIndex] = vo_triang2_mex(double(matchedPoints1), double(matchedPoints2), ...
    double(camMatrix1), double(camMatrix2));
worldPoints = cast(worldPoints, cls);
reprojectionErrors = cast(reprojectionErrors, cls);
end
