function valid_idx = isInFrontOfCamera(camMat, points3d)
% Drop-in for modules/isInFrontOfCamera.m: sign of the homogeneous depth [X 1] * camMat(:,3).
w = [points3d, ones(size(points3d, 1), 1, 'like', points3d)] * cast(camMat(:, 3), 'like', points3d);
valid_idx = w > 0;
end
