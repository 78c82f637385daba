function [curr_state, tracked_keypoints] = candidateTriangulation(prev_img, prev_state, curr_img, curr_pose, camera_params, tracker, nv)
% GPU-backed drop-in for the free function modules/candidateTriangulation.m (same signature and
% options).  Tracking and the per-candidate geometry run batched on the GPU; the order-dependent
% FIFO merge (:88-106 of the reference) is replayed on the host in candidate order.
arguments
    prev_img
    prev_state
    curr_img
    curr_pose
    camera_params
    tracker
    nv.max_temporal_recall = 50
    nv.admissible_angular_threshold = 2.5
    nv.max_landmarks = 500
    nv.max_reprojection_err = 2
    nv.maxDepth = 1000
end
[kt, ok_l] = tracker.track(prev_img, curr_img, prev_state.keypoints);
L = prev_state.landmarks(ok_l, :);
Kp = kt(ok_l, :);
[ct, ok_c] = tracker.track(prev_img, curr_img, prev_state.candidate_keypoints);
tracked_keypoints = ct(ok_c, :);
sel = find(ok_c(:).');
first = cat(3, prev_state.candidate_first_poses{sel});
[X, err, ang, front] = vo_triang_mex(double(prev_state.candidate_first_keypoints(sel, :)), double(ct(sel, :)), ...
    double(first), double(curr_pose), double(camera_params.IntrinsicMatrix));
good = front & X(:, 3) > 0 & err <= nv.max_reprojection_err;
promote = good & ang > nv.admissible_angular_threshold;
retain = good & ~promote & (prev_state.candidate_time_indxs(sel).' > -nv.max_temporal_recall);
for j = find(promote.')                      % FIFO at the cap, one candidate at a time
    if size(L, 1) >= nv.max_landmarks
        L(1, :) = []; Kp(1, :) = [];
    end
    L(end + 1, :) = X(j, :); Kp(end + 1, :) = ct(sel(j), :); %#ok<AGROW>
end
r = sel(retain.');
curr_state.landmarks = L;
curr_state.keypoints = Kp;
curr_state.candidate_keypoints = ct(r, :);
curr_state.candidate_first_keypoints = prev_state.candidate_first_keypoints(r, :);
curr_state.candidate_first_poses = prev_state.candidate_first_poses(r);
curr_state.candidate_time_indxs = prev_state.candidate_time_indxs(r) - 1;
end
