function angle = calculateAngleDeg(landmark, cam_pose0, cam_pose1)
% Drop-in for utils/calculateAngleDeg.m.  Scalar calls stay on the host (a GPU round trip per
% landmark would only add latency); the batched form lives in vo_triang_mex's third output.
r0 = landmark - cam_pose0(end, :);
r1 = landmark - cam_pose1(end, :);
c = max(-1, min(1, (r0 * r1.') / (norm(r0) * norm(r1))));
angle = acosd(c);
end
