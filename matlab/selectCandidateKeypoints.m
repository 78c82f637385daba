function new_kps_loc = selectCandidateKeypoints(img, curr_kps, nv)
% GPU-backed drop-in for modules/selectCandidateKeypoints.m (same name-value options).
% As in the reference, FilterSize is accepted but the detector runs at the toolbox default (5).
arguments
    img
    curr_kps
    nv.MinDistance double = 20
    nv.MinQuality double = 0.01
    nv.FilterSize double = 3
    nv.CandidatesToKeep double = 200
end
o = struct('MinDistance', nv.MinDistance, 'MinQuality', nv.MinQuality, 'FilterSize', 5, ...
           'CandidatesToKeep', nv.CandidatesToKeep, 'Scorer', 0, 'Kappa', 0.08);
new_kps_loc = vo_select_mex(img, curr_kps, o);
end
