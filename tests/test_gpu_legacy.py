"""The first-generation operator signatures north_star names, through HIP (SURVEY 8a rows a10 and a7's legacy variant):
the free function candidateTriangulation (modules/candidateTriangulation.m:1-117: landmark tracking inside :18-22,
`Z > 0` gate :56-57, FIFO eviction :88-91, ageing :96-105) and updateW2D3D (modules/updateW2D3D.m:9-32: the pose step
that KEEPS the MSAC inlier mask, 1 px / 10 000 trials).  Both run on images through `tracker.track` (k_lk) and the
geometry / RANSAC kernels; the checker is a sequential restatement of the reference loop on the oracle's geometry,
fed the device-tracked points.  MATLAB conventions: 1-based pixel coordinates, K in the same origin."""
import numpy as np
import pytest

from oracle import oracle, vo_frame
from visual_odometry_amd import modules, ops, synth

pytestmark = pytest.mark.gpu
K0 = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])          # zero-based origin (synth, oracle)
K1 = K0 + np.array([[0, 0, 1.0], [0, 0, 1.0], [0, 0, 0]])                   # the same camera, MATLAB's 1-based pixels


@pytest.fixture(scope="module")
def evolved():
    """A few frames of the oracle's processFrame on a 3D-consistent sequence: a state with landmarks and with
    candidates of several ages, first seen from several poses."""
    frames, poses, lift = synth.plane_sequence(8, K0)
    u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=150)
    st = vo_frame.new_state()
    st.keypoints = np.stack([u, v], 1).astype(np.float64)
    st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1])
    st.reproError = np.zeros(len(u))
    pose = None
    for k in range(1, 6):
        st, pose, status, _ = vo_frame.process_frame(frames[k - 1], frames[k], st, K0, k, maxNumLandmarks=400,
                                                     angularThreshold=2.5)
        assert status == 0
    m = modules._new_state()                                                # MATLAB conventions for modules.py
    m.landmarks = st.landmarks.copy(); m.keypoints = st.keypoints + 1.0
    m.candidate_keypoints = st.candidate_keypoints + 1.0
    m.candidate_first_keypoints = st.candidate_first_keypoints + 1.0
    m.candidate_first_poses = [p.copy() for p in st.candidate_first_poses]
    m.candidate_time_indxs = st.candidate_time_indxs.copy()
    return frames, poses, m, pose


def sequential_free_function(prev, kt, val_key, ct, val_cand, curr_pose, K, recall, angle_thr, cap, max_err):
    """modules/candidateTriangulation.m:17-116 line by line, geometry from the oracle (one candidate at a time)."""
    landmarks = list(np.asarray(prev.landmarks)[val_key]); keypoints = list(kt[val_key])              # :21-22
    ck, cfk, cfp, cti = [], [], [], []
    n_promoted = n_expired = n_dropped = 0
    for i in np.flatnonzero(val_cand):                                                                # :37
        pose0 = np.asarray(prev.candidate_first_poses[i])
        X, rep, ang, ok = oracle.triangulate(prev.candidate_first_keypoints[i][None], ct[i][None], pose0[None], curr_pose, K)
        is_valid = bool(ok[0]) and X[0, 2] > 0 and rep[0] <= max_err                                   # :56-57
        if not is_valid:                                                                              # :82-84
            continue
        if ang[0] > angle_thr:                                                                        # :86-87
            if len(landmarks) >= cap:                                                                 # :88-91 FIFO
                landmarks = landmarks[1:]; keypoints = keypoints[1:]
                n_dropped += 1
            landmarks.append(X[0]); keypoints.append(ct[i])                                           # :92-93
            n_promoted += 1
        elif prev.candidate_time_indxs[i] > -recall:                                                  # :96
            ck.append(ct[i]); cfk.append(prev.candidate_first_keypoints[i]); cfp.append(pose0)
            cti.append(prev.candidate_time_indxs[i] - 1)                                              # :97-104
        else:
            n_expired += 1
    return (np.array(landmarks).reshape(-1, 3), np.array(keypoints).reshape(-1, 2), np.array(ck).reshape(-1, 2),
            np.array(cfk).reshape(-1, 2), cfp, np.array(cti, np.int64), n_promoted, n_expired, n_dropped)


@pytest.mark.parametrize("cap,recall", [(500, 50), (120, 3)])
def test_free_candidateTriangulation_on_the_device(evolved, cap, recall):
    """Default options (:10-14) and a run where the landmark cap is hit (FIFO eviction) and old candidates expire."""
    frames, poses, prev, curr_pose = evolved
    tracker = modules.KLTTracker()
    cam = modules.CameraParams(K1)
    # the pose of the frame being entered: the true one (the function takes it as an argument, :7)
    pose6 = poses[6]
    out, tracked = modules.candidateTriangulation(frames[5], prev, frames[6], pose6, cam, tracker,
                                                  max_temporal_recall=recall, admissible_angular_threshold=2.5,
                                                  max_landmarks=cap, max_reprojection_err=2)
    kt, val_key, _ = tracker.track(frames[5], frames[6], prev.keypoints)            # the same device tracks
    ct, val_cand, _ = tracker.track(frames[5], frames[6], prev.candidate_keypoints)
    assert val_key.sum() > 50 and val_cand.sum() > 50
    lm, kp, ck, cfk, cfp, cti, n_promoted, n_expired, n_dropped = sequential_free_function(prev, kt, val_key, ct, val_cand, pose6, K1, recall, 2.5, cap, 2)
    np.testing.assert_array_equal(tracked, ct[val_cand])                            # :29
    assert not hasattr(out, "reproError")                                           # the free function keeps none
    assert len(out.landmarks) == len(lm) and len(out.candidate_keypoints) == len(ck)
    np.testing.assert_array_equal(out.keypoints, kp)                                # same tracks, same order
    np.testing.assert_allclose(out.landmarks, lm, rtol=1e-4, atol=1e-6)             # bar: 1e-4 relative
    np.testing.assert_array_equal(out.candidate_keypoints, ck)
    np.testing.assert_array_equal(out.candidate_first_keypoints, cfk)
    np.testing.assert_array_equal(out.candidate_time_indxs, cti)
    assert len(out.candidate_first_poses) == len(cfp) and all(np.array_equal(a, b) for a, b in zip(out.candidate_first_poses, cfp))
    assert n_promoted > 0
    if cap == 120:
        n0 = int(val_key.sum())
        assert n_dropped > 0 and len(lm) == n0 + n_promoted - n_dropped             # the FIFO ran (:88-91): at or above
        assert n_expired > 0                                                        # the cap one goes for each that comes
        # FIFO: the OLDEST landmarks went, the tracked ones that stay keep their order in front of the promoted
        np.testing.assert_array_equal(out.keypoints[:n0 - n_dropped], kt[val_key][n_dropped:])
    else:
        assert len(lm) == int(val_key.sum()) + n_promoted and n_expired == 0 and n_dropped == 0


def test_free_function_gates_on_world_Z(ctx):
    """:56-57 `cand_landmark(:,3) > 0` on top of triangulate's validity: a point in front of both cameras but at
    negative world Z (cameras looking down -Z) passes the class method's gate and fails the free function's."""
    R = np.diag([-1.0, 1.0, -1.0])                                  # 180 deg about y: optical axis = -Z of the world
    pose0 = synth.pose43(R, [0.0, 0, 0]); pose1 = synth.pose43(R, [1.0, 0, 0])
    X = np.array([[0.3, 0.2, -5.0], [-0.4, 0.1, -6.0], [0.1, -0.3, -4.0]])
    xy0, _ = synth.project(K1, R, np.zeros(3), X); xy1, _ = synth.project(K1, R, np.array([1.0, 0, 0]), X)
    Xg, rep, ang, ok = ops.triangulate_candidates(xy0, xy1, np.stack([pose0] * 3), pose1, K1)
    assert ok.all() and (Xg[:, 2] < 0).all() and (rep < 1e-6).all() and (ang > 2.5).all()

    class Same:
        def track(self, a, b, pts):
            pts = np.asarray(pts, np.float64).reshape(-1, 2)
            return pts.copy(), np.ones(len(pts), bool), np.ones(len(pts))
    st = modules._new_state()
    st.candidate_keypoints = xy1; st.candidate_first_keypoints = xy0
    st.candidate_first_poses = [pose0] * 3; st.candidate_time_indxs = np.array([-1, -1, -1])
    out, _ = modules.candidateTriangulation(None, st, None, pose1, modules.CameraParams(K1), Same())
    assert len(out.landmarks) == 0 and len(out.candidate_keypoints) == 0          # dropped, not aged (:82-84)
    vo = modules.VisualOdometry(modules.CameraParams(K1), (240, 320))
    vo.tracker = Same()
    cs = modules._new_state()
    out2, _ = vo.candidateTriangulation(None, st, None, cs, pose1)
    assert len(out2.landmarks) == 3                                                # the method has no such gate (:106)


def test_updateW2D3D_keeps_the_msac_inliers(evolved):
    """modules/updateW2D3D.m:9-32 on the device: tracking, P3P-MSAC at 1 px / 99 % / 10 000 trials, and the landmark
    set cut to the MSAC inliers (the live pipeline ignores that mask, modules/VisualOdometry.m:305-306)."""
    frames, poses, prev, _ = evolved
    tracker = modules.KLTTracker()
    cam = modules.CameraParams(K1)
    lm, kp, pose = modules.updateW2D3D(frames[5], frames[6], prev, cam, tracker, Seed=77)
    kt, val, _ = tracker.track(frames[5], frames[6], prev.keypoints)
    vk, vl = kt[val], prev.landmarks[val]
    # the mask that came back IS the set of points within 1 px of the returned pose (strict <, oracle arithmetic),
    # up to points sitting on the threshold
    err = oracle.reproj_error(vk, vl, pose, K1)
    inl = err < 1.0
    sure = np.abs(err - 1.0) > 1e-6
    got = np.zeros(len(vk), bool)
    got[[int(np.flatnonzero((vk == k).all(1))[0]) for k in kp]] = True
    assert (got == inl)[sure].all() and 30 < got.sum() < len(vk)                    # a real cut: some tracked points go
    np.testing.assert_array_equal(lm, vl[got]); np.testing.assert_array_equal(kp, vk[got])
    # pose: close to the truth of the sequence and a rigid motion
    np.testing.assert_allclose(pose[:3] @ pose[:3].T, np.eye(3), atol=1e-9)
    assert np.abs(pose[3] - poses[6][3]).max() < 0.3
    # identical hypotheses -> identical trial count, record and inlier set (north_star's bit-exact bar), at this
    # variant's options: the oracle's own poses are scored by the device
    o = oracle.p3p_ransac(vk, vl, K1, num_hyp=10000, max_reproj=1.0, confidence=0.99, seed=77, mode=1, min_inliers=4,
                          want_all=True, threads=4)
    g = ops.p3p_ransac(vk, vl, K1, num_hyp=10000, max_reproj_err=1.0, confidence=0.99, seed=77, mode=1, min_inliers=4,
                       hyp_poses=o["poses"])
    assert (g["status"], g["best_hyp"], g["best_sol"], g["trials"]) == (o["status"], o["best_hyp"], o["best_sol"], o["trials"])
    np.testing.assert_array_equal(g["inliers"], o["inliers"])
    assert g["trials"] < 10000                                                      # the 99 % stop was reached


def sequential_v2(prev, ct, val_cand, landmarks, keypoints, curr_pose, K, recall, angle_thr, cap, max_err):
    """modules/VisualOdometry.m:164-260 line by line: bins, one triangulation per bin and candidate against the BIN's
    pose, the three gates, promotion without eviction, ageing.  Geometry from the oracle, one candidate at a time."""
    landmarks = list(landmarks); keypoints = list(keypoints)
    last_view_id = recall + 1                                                                         # :176
    bins = {}
    for i in np.flatnonzero(val_cand):                                                                # :188-200
        view_id = int(prev.candidate_time_indxs[i]) + last_view_id
        assert 1 <= view_id <= recall
        bins.setdefault(view_id, [np.asarray(prev.candidate_first_poses[i]), []])[1].append(i)
    ck, cfk, cfp, cti = [], [], [], []
    n_promoted = n_full = n_z = 0
    for view_id in range(1, recall + 1):                                                              # :208
        if view_id not in bins:
            continue
        pose, idxs = bins[view_id]
        for idx in idxs:                                                                              # :233
            X, rep, ang, ok = oracle.triangulate(prev.candidate_first_keypoints[idx][None], ct[idx][None], pose[None], curr_pose, K)
            if bool(ok[0]) and rep[0] < max_err and not X[0, 2] > 0:
                n_z += 1
            if not (bool(ok[0]) and rep[0] < max_err and X[0, 2] > 0):                                # :229-231
                continue
            if ang[0] > angle_thr:                                                                    # :237
                if len(landmarks) < cap:                                                              # :238-241
                    landmarks.append(X[0]); keypoints.append(ct[idx]); n_promoted += 1
                else:
                    n_full += 1
            elif prev.candidate_time_indxs[idx] > -recall:                                            # :243
                ck.append(ct[idx]); cfk.append(prev.candidate_first_keypoints[idx])
                cfp.append(np.asarray(prev.candidate_first_poses[idx])); cti.append(prev.candidate_time_indxs[idx] - 1)
    return (np.array(landmarks).reshape(-1, 3), np.array(keypoints).reshape(-1, 2), np.array(ck).reshape(-1, 2),
            np.array(cfk).reshape(-1, 2), cfp, np.array(cti, np.int64), n_promoted, n_full, len(bins))


@pytest.mark.parametrize("cap", [400, 0])
def test_candidateTriangulationV2_on_the_device(evolved, cap):
    """the multi-view variant of the class (modules/VisualOdometry.m:153-262), through k_lk and the
    batched triangulation kernel, against the sequential restatement.  cap = 0 stands for "the landmark set is full":
    the cap is set to the number of landmarks the frame starts with plus three, so that promotion stops early."""
    frames, poses, prev, _ = evolved
    vo = modules.VisualOdometry(modules.CameraParams(K1), frames[0].shape, angularThreshold=2.5, maxTemporalRecall=10,
                                maxNumLandmarks=400, maxReprojectionError=2.0)
    cs = modules._new_state()
    kt, val_key, _ = vo.tracker.track(frames[5], frames[6], prev.keypoints)
    cs.landmarks = np.asarray(prev.landmarks)[val_key]; cs.keypoints = kt[val_key]
    n0 = len(cs.landmarks)
    if cap == 0:
        cap = n0 + 3
    vo.maxNumLandmarks = cap
    ct, val_cand, _ = vo.tracker.track(frames[5], frames[6], prev.candidate_keypoints)
    lm, kp, ck, cfk, cfp, cti, n_promoted, n_full, n_bins = sequential_v2(prev, ct, val_cand, cs.landmarks, cs.keypoints, poses[6], K1,
                                                                          10, 2.5, cap, 2.0)
    out, tracked = vo.candidateTriangulationV2(frames[5], prev, frames[6], cs, poses[6])
    np.testing.assert_array_equal(tracked, ct[val_cand])                            # :171
    assert n_bins >= 3                                                              # candidates of several ages
    assert len(out.landmarks) == len(lm) and len(out.candidate_keypoints) == len(ck)
    np.testing.assert_array_equal(out.keypoints, kp)                                # same tracks, the bins' order
    np.testing.assert_allclose(out.landmarks, lm, rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(out.candidate_keypoints, ck)
    np.testing.assert_array_equal(out.candidate_first_keypoints, cfk)
    np.testing.assert_array_equal(out.candidate_time_indxs, cti)
    assert len(out.candidate_first_poses) == len(cfp) and all(np.array_equal(a, b) for a, b in zip(out.candidate_first_poses, cfp))
    assert n_promoted > 0
    if cap == n0 + 3:
        assert n_promoted == 3 and n_full > 0 and len(lm) == cap                    # no eviction: the rest is dropped
    else:
        assert n_full == 0 and len(lm) == n0 + n_promoted


def test_candidateTriangulationV2_rejects_an_index_outside_the_bins():
    """bins{view_id, 1} with view_id < 1 is an indexing error in MATLAB (:190-193); here an IndexError."""
    class Same:
        def track(self, a, b, pts):
            pts = np.asarray(pts, np.float64).reshape(-1, 2)
            return pts.copy(), np.ones(len(pts), bool), np.ones(len(pts))
    vo = modules.VisualOdometry(modules.CameraParams(K1), (240, 320), maxTemporalRecall=4)
    vo.tracker = Same()
    st = modules._new_state()
    st.candidate_keypoints = np.array([[10.0, 10.0]]); st.candidate_first_keypoints = np.array([[11.0, 10.0]])
    st.candidate_first_poses = [synth.pose43(np.eye(3), [0.0, 0, 0])]; st.candidate_time_indxs = np.array([-5])
    with pytest.raises(IndexError):
        vo.candidateTriangulationV2(None, st, None, modules._new_state(), synth.pose43(np.eye(3), [1.0, 0, 0]))
