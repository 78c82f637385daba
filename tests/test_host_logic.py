"""Host-side logic of the MATLAB-signature mirror (visual_odometry_amd/modules.py) on CPU: the
order-dependent merge policies of candidateTriangulation with the device calls replaced by the oracle
(monkeypatched), so that the policy code is exercised without a GPU."""
import numpy as np
import pytest

from oracle import oracle
from visual_odometry_amd import modules, ops, synth

K = synth.KITTI_K


class FakeTracker:
    """track() returns scripted results (positions unchanged, chosen validity)."""

    def __init__(self, valid):
        self.valid = np.asarray(valid, bool)

    def track(self, img1, img2, pts):
        pts = np.asarray(pts, np.float64).reshape(-1, 2)
        if len(pts) == 0:
            return np.empty((0, 2)), np.empty((0,), bool), np.empty((0,))
        return pts.copy(), self.valid[:len(pts)].copy(), np.ones(len(pts))


@pytest.fixture
def oracle_geometry(monkeypatch):
    def tri(first_xy, curr_xy, first_poses, curr_pose, Kc, **kw):
        return oracle.triangulate(first_xy, curr_xy, np.asarray(first_poses).reshape(-1, 12), curr_pose, Kc)
    monkeypatch.setattr(ops, "triangulate_candidates", tri)


def scripted_state(n_cand=6, n_land=3):
    """Candidates 0..n-1 first seen from the origin pose; current pose 1 m to the right.  Near
    points (Z = 4) subtend a large angle, far ones (Z = 400) a tiny one."""
    pose0 = synth.pose43(np.eye(3), [0, 0, 0]); pose1 = synth.pose43(np.eye(3), [1.0, 0, 0])
    Z = np.array([4.0, 400.0, 4.0, 400.0, 4.0, 4.0])[:n_cand]
    X = np.stack([np.linspace(-1, 1, n_cand), np.zeros(n_cand), Z], 1)
    xy0, _ = synth.project(K, np.eye(3), np.zeros(3), X)
    xy1, _ = synth.project(K, np.eye(3), np.array([1.0, 0, 0]), X)
    st = modules._new_state()
    st.landmarks = np.arange(3 * n_land, dtype=np.float64).reshape(n_land, 3) + 50
    st.keypoints = np.arange(2 * n_land, dtype=np.float64).reshape(n_land, 2) + 10
    st.reproError = np.array([0.5, 1.5, 1.0])[:n_land]
    st.candidate_keypoints = xy1
    st.candidate_first_keypoints = xy0
    st.candidate_first_poses = [pose0] * n_cand
    st.candidate_time_indxs = np.array([-1, -1, -3, -10, -2, -1])[:n_cand]
    return st, pose1, X


def test_method_policy_promotes_ages_and_drops(oracle_geometry):
    """modules/VisualOdometry.m:106-139."""
    st, pose1, X = scripted_state()
    vo = modules.VisualOdometry(modules.CameraParams(K), (376, 1241), angularThreshold=1.0, maxTemporalRecall=10,
                                maxNumLandmarks=100, maxReprojectionError=2)
    vo.tracker = FakeTracker([1, 1, 1, 1, 0, 1])             # candidate 4 fails tracking: dropped (:87)
    cs = modules._new_state(); cs.landmarks = st.landmarks; cs.keypoints = st.keypoints; cs.reproError = st.reproError
    out, tracked = vo.candidateTriangulation(None, st, None, cs, pose1)
    assert len(tracked) == 5
    # near candidates 0, 2, 5 become landmarks in candidate order, appended after the old ones
    assert len(out.landmarks) == 3 + 3
    np.testing.assert_allclose(out.landmarks[3:], X[[0, 2, 5]], atol=1e-6)
    assert len(out.reproError) == 6 and (out.reproError[3:] < 1e-6).all()
    # far candidate 1 (age -1) is kept and aged; far candidate 3 (age -10 = -maxTemporalRecall) is dropped (:130)
    assert list(out.candidate_time_indxs) == [-2]
    np.testing.assert_allclose(out.candidate_keypoints, st.candidate_keypoints[[1]])
    assert len(out.candidate_first_poses) == 1


def test_method_policy_replaces_the_worst_landmark_at_the_cap(oracle_geometry):
    """:116-127: at the cap a new landmark evicts the landmark with the largest reprojection error,
    the maximum being taken before the iteration's edits (:89)."""
    st, pose1, X = scripted_state()
    vo = modules.VisualOdometry(modules.CameraParams(K), (376, 1241), maxNumLandmarks=3, maxReprojectionError=2)
    vo.tracker = FakeTracker([1, 0, 0, 0, 0, 0])
    cs = modules._new_state(); cs.landmarks = st.landmarks; cs.keypoints = st.keypoints; cs.reproError = st.reproError
    out, _ = vo.candidateTriangulation(None, st, None, cs, pose1)
    assert len(out.landmarks) == 3
    np.testing.assert_allclose(out.landmarks[:2], st.landmarks[[0, 2]])      # landmark 1 (error 1.5) evicted
    np.testing.assert_allclose(out.landmarks[2], X[0], atol=1e-6)
    assert list(np.round(out.reproError[:2], 6)) == [0.5, 1.0]


def test_free_function_policy_is_fifo(oracle_geometry):
    """modules/candidateTriangulation.m:88-93: at the cap the OLDEST landmark is dropped."""
    st, pose1, X = scripted_state()
    trk = FakeTracker([1, 1, 1, 1, 1, 1])
    out, tracked = modules.candidateTriangulation(None, st, None, pose1, modules.CameraParams(K), trk,
                                                  max_temporal_recall=10, admissible_angular_threshold=1.0,
                                                  max_landmarks=4, max_reprojection_err=2)
    assert len(tracked) == 6 and not hasattr(out, "reproError")
    # 3 old landmarks + near candidates 0, 2, 4, 5 through a 4-slot FIFO -> the last four additions survive
    np.testing.assert_allclose(out.landmarks, X[[0, 2, 4, 5]], atol=1e-6)
    assert list(out.candidate_time_indxs) == [-2]            # far candidate 1 aged, far candidate 3 expired


def test_v2_policy_bins_by_first_view_and_never_evicts(oracle_geometry):
    """modules/VisualOdometry.m:153-262: output in the bins' order (oldest view first), promotion only while there is
    room (:238-241), strict `<` on the reprojection error (:230), the ageing rule of :243."""
    st, pose1, X = scripted_state()
    vo = modules.VisualOdometry(modules.CameraParams(K), (376, 1241), angularThreshold=1.0, maxTemporalRecall=10,
                                maxNumLandmarks=5, maxReprojectionError=2)
    vo.tracker = FakeTracker([1, 1, 1, 1, 1, 1])
    cs = modules._new_state(); cs.landmarks = st.landmarks; cs.keypoints = st.keypoints
    out, tracked = vo.candidateTriangulationV2(None, st, None, cs, pose1)
    assert len(tracked) == 6
    # time indexes -1 -1 -3 -10 -2 -1 -> view ids 10 10 8 1 9 10: bins in view order hold candidates [3] [2] [4] [0 1 5].
    # Near candidates 2, 4, 0, 5 pass the angle in that order; the cap of 5 leaves room for two: 2 and 4.
    assert len(out.landmarks) == 5
    np.testing.assert_allclose(out.landmarks[:3], st.landmarks)
    np.testing.assert_allclose(out.landmarks[3:], X[[2, 4]], atol=1e-6)
    np.testing.assert_allclose(out.keypoints[3:], st.candidate_keypoints[[2, 4]])
    # far candidate 3 (age -10) expires, far candidate 1 is kept and aged; candidates 0 and 5 found no room: dropped
    assert list(out.candidate_time_indxs) == [-2]
    np.testing.assert_allclose(out.candidate_keypoints, st.candidate_keypoints[[1]])
    # the strict gate: a maximum equal to a candidate's error rejects it (the first method's `<=` accepts it)
    rep = oracle.triangulate(st.candidate_first_keypoints, st.candidate_keypoints, np.stack(st.candidate_first_poses).reshape(-1, 12), pose1, K)[1]
    vo.maxReprojectionError = float(rep[2]); vo.maxNumLandmarks = 50
    cs = modules._new_state(); cs.landmarks = st.landmarks; cs.keypoints = st.keypoints
    out2, _ = vo.candidateTriangulationV2(None, st, None, cs, pose1)
    kept = [i for i in (2, 4, 0, 5) if rep[i] < rep[2]]
    assert 2 not in kept and len(out2.landmarks) == 3 + len(kept)


def test_klt_tracker_option_checks():
    t = modules.KLTTracker()
    assert (t.NumPyramidLevels, t.MaxBidirectionalError, t.BlockSize, t.MaxIterations) == (4, 2.0, (41, 41), 50)
    for bad in ((40, 41), (3, 3), (41,)):
        with pytest.raises(ValueError):
            modules.KLTTracker(BlockSize=bad)
    with pytest.raises(ValueError):
        modules.KLTTracker(NumPyramidLevels=0)


def test_camera_params_hold_the_transposed_intrinsics():
    cp = modules.CameraParams(K)
    np.testing.assert_array_equal(cp.IntrinsicMatrix, K.T)   # utils/dataLoaderKitti.m:50-53


def test_merge_policy_closed_form_equals_the_sequential_loop():
    """The device merge for the usual sizes (k_merge_par, vo_frame.hip) does not walk the candidates one by one as
    modules/VisualOdometry.m:87-127 does; it uses two all-pairs counts.  Their equivalence to the sequential policy
    (first maximum evicted when the candidate's error is not larger, appended when below the cap), survivors'
    identity and order included, on random pools with many ties."""
    rng = np.random.default_rng(0)

    def sequential(init, cand, cap):
        vals = list(init); ids = [("L", i) for i in range(len(init))]; acc = []
        for j, r in enumerate(cand):
            mi, mx = (int(np.argmax(vals)), max(vals)) if vals else (-1, -np.inf)       # :89-90
            if len(vals) >= cap and r <= mx:                                            # :116-122
                del vals[mi]; del ids[mi]
            if len(vals) < cap:                                                         # :123-127
                vals.append(r); ids.append(("C", j)); acc.append(j)
        return ids, acc

    def closed(init, cand, cap):
        n0 = len(init); pool = np.concatenate([init, cand]).astype(float); order = np.arange(len(pool))
        rank = (pool[None, :] < pool[:, None]).sum(1) + ((pool[None, :] == pool[:, None]) & (order[None, :] > order[:, None])).sum(1)
        acc = [j for j in range(len(cand)) if (pool[:n0 + j] < cand[j]).sum() < cap]
        surv = rank < cap
        return [("L", i) for i in range(n0) if surv[i]] + [("C", j) for j in acc if surv[n0 + j]], acc
    for _ in range(1500):
        cap = int(rng.integers(1, 14)); n0 = min(int(rng.integers(0, 12)), cap); q = int(rng.integers(1, 6))
        init = rng.integers(0, q, n0) / 2.0; cand = rng.integers(0, q, int(rng.integers(0, 15))) / 2.0
        assert sequential(init, cand, cap) == closed(init, cand, cap)


def test_sort_based_merge_equals_the_sequential_loop_on_tie_heavy_pools():
    """The device merge for LARGE pools (round 3: k_sort_tiles / k_sort_decide, vo_frame.hip) takes both
    decisions from one stable merge sort by reproError: `before` = the strictly smaller keys of the left-hand runs an
    entry is merged with, level by level; `rank` = the sorted position mirrored inside its group of equal keys.  The
    same steps in numpy (same run structure, same binary searches, a padded tail of +inf keys, partner runs shorter
    than their neighbours) against the sequential policy of modules/VisualOdometry.m:87-127, on pools with many ties.
    tests/test_gpu_merge.py holds the kernels themselves to the same policy up to 60 000 entries."""
    rng = np.random.default_rng(1)

    def sequential(init, cand, cap):
        vals = list(init); ids = list(range(len(init))); acc = np.zeros(len(cand), bool)
        for j, r in enumerate(cand):
            mi, mx = (int(np.argmax(vals)), max(vals)) if vals else (-1, -np.inf)       # :89-90
            if len(vals) >= cap and r <= mx:                                            # :116-122
                del vals[mi]; del ids[mi]
            if len(vals) < cap:                                                         # :123-127
                vals.append(r); ids.append(len(init) + j); acc[j] = True
        surv = np.zeros(len(init) + len(cand), bool); surv[ids] = True
        return surv, acc

    def by_sort(init, cand, cap, tile):
        pool = np.concatenate([init, cand]).astype(float); P = len(pool)
        padded = (P + tile - 1) // tile * tile
        key = np.full(padded, np.inf); key[:P] = pool; idx = np.arange(padded); before = np.zeros(padded, int)
        L = 1
        while L < padded:
            nk, ni = key.copy(), idx.copy()
            for p in range(padded):
                run = p // L; base = (run >> 1) * 2 * L; v = key[p]
                if base + L >= padded:
                    d, sm = p, 0                                   # a last run without a partner moves as it is
                elif run & 1:                                      # right-hand run: behind the left keys <= v
                    le = np.searchsorted(key[base:base + L], v, "right")
                    sm = np.searchsorted(key[base:base + le], v, "left")
                    d = base + (p - (base + L)) + le
                else:                                              # left-hand run: behind the right keys < v
                    sm = 0
                    d = base + (p - base) + np.searchsorted(key[base + L:min(base + 2 * L, padded)], v, "left")
                nk[d] = v; ni[d] = idx[p]; before[idx[p]] += sm
            key, idx = nk, ni; L *= 2
        surv = np.zeros(P, bool); acc = np.zeros(P, bool)
        for p in range(P):
            lo = np.searchsorted(key[:P], key[p], "left"); hi = np.searchsorted(key[:P], key[p], "right")
            surv[idx[p]] = lo + (hi - 1 - p) < cap
            acc[idx[p]] = before[idx[p]] < cap
        return surv, acc[len(init):]
    for _ in range(300):
        cap = int(rng.integers(1, 30)); n0 = min(int(rng.integers(0, 28)), cap); q = int(rng.integers(1, 6))
        init = rng.integers(0, q, n0) / 2.0; cand = rng.integers(0, q, int(rng.integers(0, 40))) / 2.0
        ws, wa = sequential(init, cand, cap)
        gs, ga = by_sort(init, cand, cap, tile=int(rng.choice([1, 4, 8, 16])))
        assert np.array_equal(ga, wa)
        assert np.array_equal(gs[:n0], ws[:n0]) and np.array_equal(gs[n0:] & ga, ws[n0:])
