"""Host-side mirror of the reference's operator interface for the hot path (same names, argument
meaning and error behaviour as the MATLAB originals), over the C ABI of libvo_hip.so.

MATLAB/Octave are absent from the build image, so the host side above the C ABI is written in Python
and mirrors the MATLAB signatures; the MEX gateways a maintainer would compile live in `mex/` and the
MATLAB replacement files in `matlab/`.  Conventions follow MATLAB: points are
N x 2 `[x y]` with 1-based pixel coordinates, poses are 4 x 3 `[R_WC; t_WC]`
(modules/VisualOdometry.m:318) and `cameraParams.IntrinsicMatrix` is K^T
(utils/dataLoaderKitti.m:50-53).  A Fortran-ordered image is passed through as MATLAB memory.

    KLTTracker                  modules/KLTTracker.m:1-43
    selectCandidateKeypoints    modules/selectCandidateKeypoints.m:1-36
    candidateTriangulation      modules/candidateTriangulation.m:1-117 (free function)
    updateW2D3D                 modules/updateW2D3D.m:1-46 (legacy pose step that keeps the MSAC inliers)
    VisualOdometry              modules/VisualOdometry.m (processFrame :264-346, candidateTriangulation :63-151)
    estimateWorldCameraPose     the toolbox call of modules/VisualOdometry.m:285-289
    calculateAngleDeg           utils/calculateAngleDeg.m:1-11
    isInFrontOfCamera           modules/isInFrontOfCamera.m:1-12
    harris, selectKeypoints, p3p  old/harris.m, old/selectKeypoints.m, old/p3p.m
    bootstrap, estimateEssentialMatrix, relativeCameraPose   modules/bootstrap.m:1-135 and its toolbox calls
"""
from types import SimpleNamespace

import numpy as np

from . import ops


class CameraParams(SimpleNamespace):
    """Stand-in for MATLAB's cameraParameters: `.IntrinsicMatrix` = K^T (utils/dataLoaderKitti.m:53)."""

    def __init__(self, K):
        K = np.asarray(K, np.float64)
        super().__init__(IntrinsicMatrix=K.T.copy(), K=K.copy())


class KLTTracker:
    """modules/KLTTracker.m:1-43.  Options and defaults of :11-14."""

    def __init__(self, NumPyramidLevels=4, MaxBidirectionalError=2, BlockSize=(41, 41), MaxIterations=50):
        bs = tuple(int(b) for b in np.ravel(BlockSize))
        if len(bs) != 2 or any(b < 5 or b % 2 == 0 for b in bs):
            # vision.PointTracker: "BlockSize must be odd integers >= 5"
            raise ValueError("BlockSize must be two odd integers >= 5")
        if int(NumPyramidLevels) < 1 or int(MaxIterations) < 1:
            raise ValueError("NumPyramidLevels and MaxIterations must be >= 1")
        self.NumPyramidLevels = int(NumPyramidLevels)
        self.MaxBidirectionalError = float(MaxBidirectionalError)
        self.BlockSize = bs
        self.MaxIterations = int(MaxIterations)

    def track(self, img1, img2, img1_pts):
        """[points, validity, scores] = track(obj, img1, img2, img1_pts)  (modules/KLTTracker.m:22-41).
        Empty input returns typed empties (:33-36)."""
        return ops.klt_track(img1, img2, img1_pts, levels=self.NumPyramidLevels,
                             max_bidir_err=self.MaxBidirectionalError, block=self.BlockSize,
                             max_iters=self.MaxIterations)


def selectCandidateKeypoints(img, curr_kps, MinDistance=20, MinQuality=0.01, FilterSize=3, CandidatesToKeep=200,
                             Scorer="mineigen"):
    """modules/selectCandidateKeypoints.m:1-36 (defaults :13-16).  As in the reference, FilterSize is
    accepted but NOT forwarded to the detector (:20), which runs at the toolbox default of 5."""
    del FilterSize
    return ops.select_candidates(img, curr_kps, min_distance=MinDistance, min_quality=MinQuality, filter_size=5,
                                 keep=int(CandidatesToKeep), scorer=Scorer)


def calculateAngleDeg(landmark, cam_pose0, cam_pose1):
    """utils/calculateAngleDeg.m:1-11 (1 x 3 landmark -> scalar; N x 3 -> N)."""
    lm = np.asarray(landmark, np.float64)
    a = ops.angle_deg(lm.reshape(-1, 3), cam_pose0, cam_pose1)
    return float(a[0]) if lm.ndim == 1 or lm.shape[0] == 1 else a


def isInFrontOfCamera(camMat, points3d):
    """modules/isInFrontOfCamera.m:1-12: camMat 4 x 3, points N x 3 -> N logical."""
    return ops.in_front(camMat, points3d)


def triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2):
    """The toolbox call at modules/VisualOdometry.m:97-101 / modules/candidateTriangulation.m:47-51 / processFrame.m:55-59:
    N x 2 points, 4 x 3 camera matrices -> (worldPoints N x 3, reprojectionErrors N, validIndex N); outputs follow the
    class of the points (matlab/triangulate.m does the same)."""
    p1 = np.asarray(matchedPoints1)
    dt = np.float32 if p1.dtype == np.float32 else np.float64
    X, rep, valid = ops.triangulate(p1, matchedPoints2, camMatrix1, camMatrix2)
    return X.astype(dt), rep.astype(dt), valid


def estimateWorldCameraPose(imagePoints, worldPoints, cameraParams, MaxNumTrials=1000, Confidence=99,
                            MaxReprojectionError=1, Seed=1023, SampleIdx=None):
    """The pose step's toolbox call (modules/VisualOdometry.m:285-289): P3P inside MSAC.
    Returns (worldOrientation 3x3, worldLocation 1x3, inlierIdx, status) with status 0 ok /
    1 not enough points / 2 not enough inliers.  Confidence is a percentage, as in MATLAB."""
    K = cameraParams.K if hasattr(cameraParams, "K") else np.asarray(cameraParams.IntrinsicMatrix).T
    # MATLAB pixel coordinates and K share the 1-based convention: no shift is applied
    r = ops.p3p_ransac(imagePoints, worldPoints, K, num_hyp=MaxNumTrials, max_reproj_err=MaxReprojectionError,
                       confidence=Confidence / 100.0, seed=Seed, mode=1, min_inliers=4, sample_idx=SampleIdx)
    return r["R"], r["t"].reshape(1, 3), r["inliers"], r["status"]


def harris(img, patch_size, kappa):
    """old/harris.m:1-25."""
    return ops.corner_response(img, "harris", patch_size, kappa)


def selectKeypoints(scores, num, r):
    """old/selectKeypoints.m:1-16 -> 2 x num [row; col], 1-based."""
    return ops.select_keypoints_greedy(scores, num, r)


def p3p(worldPoints, imageVectors):
    """old/p3p.m:41-217: 3 x 3 columns of world points / unit bearings -> 3 x 16 = four [C R] blocks."""
    P = np.asarray(worldPoints, np.float64).T  # rows = points
    f = np.asarray(imageVectors, np.float64).T
    o = ops.p3p_batch(P[None], f[None])[0]
    out = np.zeros((3, 16))
    for s in range(4):
        out[:, 4 * s] = o[s, :3]
        out[:, 4 * s + 1:4 * s + 4] = o[s, 3:].reshape(3, 3)
    return out


def _new_state():
    return SimpleNamespace(landmarks=np.zeros((0, 3)), keypoints=np.zeros((0, 2)), reproError=np.zeros((0,)),
                           candidate_keypoints=np.zeros((0, 2)), candidate_first_keypoints=np.zeros((0, 2)),
                           candidate_first_poses=[], candidate_time_indxs=np.zeros((0,), np.int64))


def _triangulate_tracked(prev_state, candidate_tracked, val_cand, curr_pose, K):
    """One batched call for every validly tracked candidate (the loop body :93-113 up to the gates)."""
    idx = np.flatnonzero(val_cand)
    if idx.size == 0:
        return idx, np.zeros((0, 3)), np.zeros(0), np.zeros(0), np.zeros(0, bool)
    poses = np.stack([np.asarray(prev_state.candidate_first_poses[i], np.float64) for i in idx])
    X, rep, ang, ok = ops.triangulate_candidates(np.asarray(prev_state.candidate_first_keypoints)[idx],
                                                 np.asarray(candidate_tracked)[idx], poses, curr_pose, K)
    return idx, X, rep, ang, ok


def candidateTriangulation(prev_img, prev_state, curr_img, curr_pose, camera_params, tracker,
                           max_temporal_recall=50, admissible_angular_threshold=2.5, max_landmarks=500,
                           max_reprojection_err=2, maxDepth=1000):
    """modules/candidateTriangulation.m:1-117 (free function, first-generation policy: FIFO eviction
    at the cap :88-91, extra Z > 0 gate :56).  Geometry runs batched on the GPU; the order-dependent
    merge is applied in candidate order on the host."""
    del maxDepth
    K = camera_params.K
    kt, val_key, _ = tracker.track(prev_img, curr_img, prev_state.keypoints)
    landmarks = np.asarray(prev_state.landmarks)[val_key]
    keypoints = kt[val_key]
    ct, val_cand, _ = tracker.track(prev_img, curr_img, prev_state.candidate_keypoints)
    tracked_keypoints = ct[val_cand]
    cs = _new_state()
    idx, X, rep, ang, ok = _triangulate_tracked(prev_state, ct, val_cand, curr_pose, K)
    lm, kp = list(landmarks), list(keypoints)
    ck, cfk, cfp, cti = [], [], [], []
    for j, i in enumerate(idx):
        if not (ok[j] and X[j, 2] > 0 and rep[j] <= max_reprojection_err):
            continue
        if ang[j] > admissible_angular_threshold:
            if len(lm) >= max_landmarks:
                lm = lm[1:]; kp = kp[1:]
            lm.append(X[j]); kp.append(ct[i])
        elif prev_state.candidate_time_indxs[i] > -max_temporal_recall:
            ck.append(ct[i]); cfk.append(prev_state.candidate_first_keypoints[i])
            cfp.append(prev_state.candidate_first_poses[i]); cti.append(prev_state.candidate_time_indxs[i] - 1)
    cs.landmarks = np.array(lm).reshape(-1, 3); cs.keypoints = np.array(kp).reshape(-1, 2)
    cs.candidate_keypoints = np.array(ck).reshape(-1, 2); cs.candidate_first_keypoints = np.array(cfk).reshape(-1, 2)
    cs.candidate_first_poses = cfp; cs.candidate_time_indxs = np.array(cti, np.int64)
    del cs.reproError
    return cs, tracked_keypoints


def updateW2D3D(prev_img, curr_img, prev_state, camera_parameters, tracker, Seed=1023):
    """[landmarks, keypoints, curr_pose] = updateW2D3D(prev_img, curr_img, prev_state, camera_parameters, tracker)
    (modules/updateW2D3D.m:1-46): the first-generation pose step.  Track the landmarks' keypoints (:9-11), pose by
    estimateWorldCameraPose with MaxReprojectionError 1, Confidence 99, MaxNumTrials 10000 (:17-22), and -- unlike
    VisualOdometry.processFrame -- keep exactly the MSAC inliers among the tracked points (:26-30).  The figures of
    :13-14,34-42 are not drawn."""
    curr_pts, val_idx, _ = tracker.track(prev_img, curr_img, prev_state.keypoints)
    R_WC, T_WC, inl_idx, _ = estimateWorldCameraPose(np.asarray(curr_pts, np.float64)[val_idx],
                                                     np.asarray(prev_state.landmarks, np.float64)[val_idx],
                                                     camera_parameters, MaxReprojectionError=1, Confidence=99,
                                                     MaxNumTrials=10000, Seed=Seed)
    val = np.flatnonzero(val_idx)
    keep = val[np.flatnonzero(inl_idx)]
    return np.asarray(prev_state.landmarks)[keep], np.asarray(curr_pts)[keep], np.vstack([R_WC, T_WC])


class VisualOdometry:
    """modules/VisualOdometry.m: constructor defaults :22-25, tracker :33-37."""

    def __init__(self, cameraParams, imageSize, angularThreshold=1.0, maxTemporalRecall=10, maxNumLandmarks=300,
                 maxReprojectionError=3, Seed=1023):
        self.cameraParams = cameraParams
        self.imageSize = imageSize
        self.angularThreshold = float(angularThreshold)
        self.maxTemporalRecall = int(maxTemporalRecall)
        self.maxNumLandmarks = int(maxNumLandmarks)
        self.maxReprojectionError = float(maxReprojectionError)
        self.tracker = KLTTracker(NumPyramidLevels=4, MaxBidirectionalError=2, BlockSize=(41, 41), MaxIterations=50)
        self.seed = int(Seed)
        self.frame = 0

    def computeReprojectionError(self, landmarks, keypoints, cameraPose):
        """modules/VisualOdometry.m:48-61."""
        return ops.reproj_error(keypoints, landmarks, cameraPose, self.cameraParams.K)

    def candidateTriangulation(self, prev_img, prev_state, curr_img, curr_state, curr_pose):
        """modules/VisualOdometry.m:63-151: replace-worst policy at the cap (:116-122), ageing (:130-139)."""
        landmarks = list(np.asarray(curr_state.landmarks).reshape(-1, 3))
        keypoints = list(np.asarray(curr_state.keypoints).reshape(-1, 2))
        repro = list(np.asarray(curr_state.reproError).reshape(-1))
        ct, val_cand, _ = self.tracker.track(prev_img, curr_img, prev_state.candidate_keypoints)
        tracked_keypoints = ct[val_cand]
        idx, X, rep, ang, ok = _triangulate_tracked(prev_state, ct, val_cand, curr_pose, self.cameraParams.K)
        ck, cfk, cfp, cti = [], [], [], []
        for j, i in enumerate(idx):
            # :89-90 maximum BEFORE this iteration's edits (first maximum wins)
            if repro:
                mi = int(np.argmax(repro)); mx = repro[mi]
            else:
                mi, mx = -1, -np.inf
            if not (ok[j] and rep[j] <= self.maxReprojectionError):  # :106-111
                continue
            if ang[j] > self.angularThreshold:                        # :113
                if len(landmarks) >= self.maxNumLandmarks and rep[j] <= mx:  # :116-122
                    del landmarks[mi]; del keypoints[mi]; del repro[mi]
                if len(landmarks) < self.maxNumLandmarks:             # :123-127
                    landmarks.append(X[j]); keypoints.append(ct[i]); repro.append(rep[j])
            elif prev_state.candidate_time_indxs[i] > -self.maxTemporalRecall:  # :130-139
                ck.append(ct[i]); cfk.append(prev_state.candidate_first_keypoints[i])
                cfp.append(prev_state.candidate_first_poses[i]); cti.append(prev_state.candidate_time_indxs[i] - 1)
        curr_state.landmarks = np.array(landmarks).reshape(-1, 3)
        curr_state.keypoints = np.array(keypoints).reshape(-1, 2)
        curr_state.reproError = np.array(repro).reshape(-1)
        curr_state.candidate_keypoints = np.array(ck).reshape(-1, 2)
        curr_state.candidate_first_keypoints = np.array(cfk).reshape(-1, 2)
        curr_state.candidate_first_poses = cfp
        curr_state.candidate_time_indxs = np.array(cti, np.int64)
        return curr_state, tracked_keypoints

    def candidateTriangulationV2(self, prev_img, prev_state, curr_img, curr_state, curr_pose):
        """modules/VisualOdometry.m:153-262 (never called by the reference).  The candidates are
        binned by the view they were first seen in -- view id = time index + maxTemporalRecall + 1 (:176,190-200), the
        bin's pose is the first pose filed in it (:194-196) -- and every non-empty bin, in view order, is triangulated
        against the current view (:205-228).  Gates: valid, reprojection error STRICTLY below the maximum, world Z > 0
        (:229-231); a candidate past the angle threshold is promoted while there is room and dropped otherwise -- no
        eviction, no reproError bookkeeping (:238-241); the others age as in the first method (:243-252).
        All bins go through the device in ONE batched call (each candidate carries its bin's pose); the merge runs on
        the host in the reference's order: bins by view id, candidates by index inside a bin."""
        landmarks = list(np.asarray(curr_state.landmarks).reshape(-1, 3))
        keypoints = list(np.asarray(curr_state.keypoints).reshape(-1, 2))
        ct, val_cand, _ = self.tracker.track(prev_img, curr_img, prev_state.candidate_keypoints)   # :169-171
        tracked_keypoints = ct[val_cand]
        last_view_id = self.maxTemporalRecall + 1                                                  # :176
        bins = [[] for _ in range(self.maxTemporalRecall)]
        bin_pose = [None] * self.maxTemporalRecall
        for i in np.flatnonzero(val_cand):                                                         # :188-200
            view_id = int(prev_state.candidate_time_indxs[i]) + last_view_id
            if not 1 <= view_id <= self.maxTemporalRecall:     # MATLAB: index out of bounds on bins{view_id, 1}
                raise IndexError("candidateTriangulationV2: candidate_time_indxs outside [-maxTemporalRecall, -1]")
            if not bins[view_id - 1]:
                bin_pose[view_id - 1] = np.asarray(prev_state.candidate_first_poses[i], np.float64)
            bins[view_id - 1].append(int(i))
        order = [i for b in bins for i in b]
        ck, cfk, cfp, cti = [], [], [], []
        if order:
            poses0 = np.stack([bin_pose[b] for b, m in enumerate(bins) for _ in m])
            X, rep, ang, ok = ops.triangulate_candidates(np.asarray(prev_state.candidate_first_keypoints)[order],
                                                         np.asarray(ct)[order], poses0, curr_pose, self.cameraParams.K)
            for j, i in enumerate(order):
                if not (ok[j] and rep[j] < self.maxReprojectionError and X[j, 2] > 0):             # :229-231
                    continue
                if ang[j] > self.angularThreshold:                                                 # :237
                    if len(landmarks) < self.maxNumLandmarks:                                      # :238-241
                        landmarks.append(X[j]); keypoints.append(ct[i])
                elif prev_state.candidate_time_indxs[i] > -self.maxTemporalRecall:                 # :243-252
                    ck.append(ct[i]); cfk.append(prev_state.candidate_first_keypoints[i])
                    cfp.append(prev_state.candidate_first_poses[i]); cti.append(prev_state.candidate_time_indxs[i] - 1)
        curr_state.landmarks = np.array(landmarks).reshape(-1, 3)
        curr_state.keypoints = np.array(keypoints).reshape(-1, 2)
        curr_state.candidate_keypoints = np.array(ck).reshape(-1, 2)
        curr_state.candidate_first_keypoints = np.array(cfk).reshape(-1, 2)
        curr_state.candidate_first_poses = cfp
        curr_state.candidate_time_indxs = np.array(cti, np.int64)
        return curr_state, tracked_keypoints

    def processFrame(self, prev_img, curr_img, prev_state):
        """[curr_state, curr_pose, pose_status] = processFrame(obj, prev_img, curr_img, prev_state)
        (modules/VisualOdometry.m:264-346)."""
        tk, val_idx, _ = self.tracker.track(prev_img, curr_img, prev_state.keypoints)           # :278-281
        vk = tk[val_idx]
        vl = np.asarray(prev_state.landmarks)[val_idx]
        self.frame += 1
        R, T, _inl, status = estimateWorldCameraPose(vk.astype(np.float64), vl.astype(np.float64),  # :285-289
                                                     self.cameraParams, MaxNumTrials=5000, Confidence=99,
                                                     MaxReprojectionError=2, Seed=self.seed + self.frame)
        curr_pose = np.vstack([R, T])                                                            # :318
        err = self.computeReprojectionError(vl, vk, curr_pose) if len(vk) else np.zeros(0)       # :298-300
        keep = err < self.maxReprojectionError                                                   # :302
        cs = _new_state()
        cs.keypoints = vk[keep]; cs.landmarks = vl[keep]; cs.reproError = err[keep]              # :308-315
        cs, _ = self.candidateTriangulation(prev_img, prev_state, curr_img, cs, curr_pose)       # :322
        both = np.vstack([cs.keypoints.reshape(-1, 2), cs.candidate_keypoints.reshape(-1, 2)])
        new = selectCandidateKeypoints(curr_img, both, MinQuality=0.001, FilterSize=5, MinDistance=7,
                                       CandidatesToKeep=100)                                     # :330-335
        cs.candidate_keypoints = np.vstack([cs.candidate_keypoints.reshape(-1, 2), new])         # :338-345
        cs.candidate_first_keypoints = np.vstack([cs.candidate_first_keypoints.reshape(-1, 2), new])
        cs.candidate_first_poses = list(cs.candidate_first_poses) + [curr_pose] * len(new)
        cs.candidate_time_indxs = np.concatenate([cs.candidate_time_indxs, -np.ones(len(new), np.int64)])
        return cs, curr_pose, status


# ---------------------------------------------------------------- bootstrap (modules/bootstrap.m)
def _essential(m0, m1, K, solver, **kw):
    """the relative-pose step by either solver: 'five-point' (Nister; what the toolbox function is built around, works on
    planar scenes; definition N10b) or 'eight-point' (definition N10)"""
    if solver == "five-point":
        return ops.essential_ransac_five(m0, m1, K, min_inliers=5, **kw)
    if solver == "eight-point":
        return ops.essential_ransac(m0, m1, K, min_inliers=8, **kw)
    raise ValueError("Solver must be 'five-point' or 'eight-point'")


def estimateEssentialMatrix(matchedPoints1, matchedPoints2, cameraParams, MaxNumTrials=500, Confidence=99,
                            MaxDistance=0.1, Seed=1023, Solver="five-point"):
    """[E, inliersIndex, status] of the toolbox call at modules/bootstrap.m:76-79: MSAC around a five-point solver
    (definition N10b; Solver='eight-point': the normalised eight-point algorithm, N10).  E satisfies x2' E x1 = 0 on
    normalised points."""
    K = cameraParams.K if hasattr(cameraParams, "K") else np.asarray(cameraParams.IntrinsicMatrix).T
    r = _essential(matchedPoints1, matchedPoints2, K, Solver, num_hyp=MaxNumTrials, max_distance=MaxDistance,
                   confidence=Confidence / 100.0, seed=Seed)
    return r["E"], r["inliers"], r["status"]


def _decompose_essential(E):
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    Ra, Rb, t = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    return [(Ra, t), (Ra, -t), (Rb, t), (Rb, -t)]


def relativeCameraPose(E, cameraParams, inlierPoints1, inlierPoints2):
    """[orientation, location, validPointsFraction] (modules/bootstrap.m:93-94): of the four
    decompositions of E the one with most points in front of both cameras (batched GPU triangulation)."""
    K = cameraParams.K if hasattr(cameraParams, "K") else np.asarray(cameraParams.IntrinsicMatrix).T
    p0 = np.asarray(inlierPoints1, np.float64); p1 = np.asarray(inlierPoints2, np.float64)
    n = len(p0)
    pose0 = np.vstack([np.eye(3), np.zeros((1, 3))])
    best = None
    for R, t in _decompose_essential(np.asarray(E, np.float64)):
        pose1 = np.vstack([R, (-(R.T @ t)).reshape(1, 3)])
        _, _, _, ok = ops.triangulate_candidates(p0, p1, pose0[None], pose1, K, first_pose_idx=np.ones(n, np.int32))
        c = int(ok.sum())
        if best is None or c > best[0]:
            best = (c, pose1)
    return best[1][:3], best[1][3].reshape(1, 3), best[0] / max(n, 1)


def bootstrap(img0, img1, cameraParams, PlotResult=False, MinDepth=0.0, MaxDepth=5000.0, MinNumLandmarks=50,
              FeatureMatchingMode="KLT", MinQuality=0.01, FilterSize=5, uniformFeaturesPercentage=100.0,
              Seed=1023, MaxAttempts=20, return_info=False, Solver="five-point"):
    """[keypoints, landmarks, reproError, pose] = bootstrap(img0, img1, cameraParams, ...)
    (modules/bootstrap.m:1-135).  'HardMatching' (:40-55, descriptor matching) is not built: only the
    default KLT mode.  The reference loops until enough landmarks exist (:74); here at most MaxAttempts
    reseeded attempts, then an error.  Solver: the minimal solver inside estimateEssentialMatrix (:76-79), 'five-point'
    (default) or 'eight-point'."""
    del PlotResult
    if FeatureMatchingMode != "KLT":
        raise NotImplementedError("bootstrap: only FeatureMatchingMode='KLT' is built")
    K = cameraParams.K if hasattr(cameraParams, "K") else np.asarray(cameraParams.IntrinsicMatrix).T
    H, W = np.asarray(img0).shape
    cap = H * W // 4 + 16
    none = np.zeros((0, 2), np.float32)
    pts0 = ops.select_candidates(img0, none, min_distance=0.0, min_quality=MinQuality, filter_size=FilterSize, keep=cap)  # :30-32
    n_keep = int(round(len(pts0) * uniformFeaturesPercentage / 100.0))                                                    # :36-38
    if n_keep < len(pts0):
        pts0 = ops.select_candidates(img0, none, min_distance=0.0, min_quality=MinQuality, filter_size=FilterSize, keep=n_keep)
    points, validity, _ = KLTTracker().track(img0, img1, pts0.astype(np.float64))                                       # :57-58
    m0 = pts0.astype(np.float64)[validity]; m1 = points[validity]                                                        # :59-60
    pose0 = np.vstack([np.eye(3), np.zeros((1, 3))])
    for attempt in range(int(MaxAttempts)):                                                                               # :74
        r = _essential(m0, m1, K, Solver, num_hyp=2000, max_distance=0.09, confidence=0.995, seed=Seed + attempt)       # :76-79
        inl = r["inliers"]
        if r["status"] != 0 or inl.sum() < MinNumLandmarks:                                                               # :83-86
            continue
        a, b = m0[inl], m1[inl]
        R1, t1, frac = relativeCameraPose(r["E"], cameraParams, a, b)                                                     # :93-94
        if inl.sum() * frac < MinNumLandmarks or not np.isfinite(R1).all() or not np.isfinite(t1).all():                  # :97-100
            continue
        pose = np.vstack([R1, t1])
        X, rep, _, ok = ops.triangulate_candidates(a, b, pose0[None], pose, K, first_pose_idx=np.ones(len(a), np.int32))  # :120-121
        ok = ok & (X[:, 2] > MinDepth) & (X[:, 2] <= MaxDepth)                                                            # :124-125
        if ok.sum() >= MinNumLandmarks:                                                                                   # :128
            out = (b[ok], X[ok], rep[ok], pose)                                                                           # :132-135
            if return_info:
                return out + (dict(corners=len(pts0), tracked=int(validity.sum()), inliers=int(inl.sum()),
                                   attempts=attempt + 1, valid_fraction=frac, trials=r["trials"]),)
            return out
    raise RuntimeError("bootstrap: no attempt produced enough landmarks")
