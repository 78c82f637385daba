"""CPU checks that pin the oracle itself (no GPU): analytic known answers and independent numpy /
scipy restatements of the reference's OPEN code (old/harris.m, old/selectKeypoints.m, old/p3p.m,
old/solveQuartic.m, utils/calculateAngleDeg.m, modules/isInFrontOfCamera.m).  The reference ships
no golden vectors or numeric tests: parity is unpinned beyond these.
"""
import numpy as np
import pytest
from scipy import signal

from oracle import oracle
from visual_odometry_amd import synth


# ---------------------------------------------------------------- KLT (N1-N5)
def test_pyramid_shapes_and_constant_image():
    img = np.full((37, 53), 77, np.uint8)
    pyr = oracle.build_pyramid(img, 4)
    assert [p.shape for p in pyr] == [(37, 53), (19, 27), (10, 14), (5, 7)]
    for p in pyr:
        assert np.all(p == 77.0)  # binomial weights sum to one exactly


def test_gradients_of_a_ramp():
    v, u = np.mgrid[0:20, 0:30].astype(np.float64)
    gu, gv = oracle.gradients(3.0 * u + 5.0 * v)
    assert np.allclose(gu[1:-1, 1:-1], 3.0) and np.allclose(gv[1:-1, 1:-1], 5.0)


@pytest.mark.parametrize("shift", [(0.37, -0.81), (3.3, -2.6), (-5.5, 4.25)])
def test_klt_recovers_a_known_translation(shift):
    """band-limited texture translated by a known sub-pixel shift."""
    img1, img2, truth = synth.frame_pair(nv=160, nu=200, seed=5, shift=shift)
    pu, pv = synth.keypoints(60, nv=160, nu=200, seed=6, border=30)
    qu, qv, valid, score, fb, iters = oracle.klt_track(img1, img2, pu, pv)
    tu, tv = truth(pu, pv)
    err = np.hypot(qu - tu, qv - tv)
    assert valid.all()
    assert err.max() < 0.05 and np.median(err) < 0.02  # uint8 quantisation limits accuracy, not LK
    assert (score > 0.97).all() and (fb < 0.05).all()


def test_klt_out_of_bounds_and_flat_points_are_invalid():
    img1, img2, _ = synth.frame_pair(nv=120, nu=160, seed=7, shift=(30.0, 0.0))
    # the true motion carries the first point outside the frame; the second sits on a flat patch
    flat1 = img1.copy(); flat1[40:100, 40:100] = 128
    flat2 = img2.copy(); flat2[40:100, 40:100] = 128
    qu, qv, valid, score, fb, it = oracle.klt_track(flat1, flat2, np.array([150.0, 70.0]), np.array([60.0, 70.0]),
                                                    levels=1, wu=21, wv=21)
    assert not valid[1]
    qu, qv, valid, score, fb, it = oracle.klt_track(img1, img2, np.array([152.0]), np.array([60.0]), levels=3)
    assert not valid[0]


def test_klt_empty_input():
    img = np.zeros((32, 32), np.uint8)
    qu, qv, valid, score, fb, it = oracle.klt_track(img, img, np.zeros(0), np.zeros(0))
    assert qu.size == 0 and valid.size == 0 and score.size == 0


def test_klt_rejects_even_windows():
    img = np.zeros((32, 32), np.uint8)
    with pytest.raises(ValueError):
        oracle.klt_track(img, img, np.ones(1), np.ones(1), wu=20, wv=21)


# ---------------------------------------------------------------- selection (N6-N8)
def harris_numpy(img, patch, kappa):
    """old/harris.m:1-25 restated with scipy (img is indexed [row, col])."""
    img = img.astype(np.float64)
    para = np.array([[-1.0, 0.0, 1.0]]); orth = np.array([[1.0, 2.0, 1.0]])
    Ix = signal.convolve2d(signal.convolve2d(img, orth.T, "valid"), para, "valid")   # :6
    Iy = signal.convolve2d(signal.convolve2d(img, para.T, "valid"), orth, "valid")   # :7
    ones = np.ones((patch, patch))
    sxx = signal.convolve2d(Ix ** 2, ones, "valid"); syy = signal.convolve2d(Iy ** 2, ones, "valid")
    sxy = signal.convolve2d(Ix * Iy, ones, "valid")
    sc = (sxx * syy - sxy ** 2) - kappa * (sxx + syy) ** 2                           # :18-19
    sc[sc < 0] = 0                                                                    # :21
    pr = patch // 2
    return np.pad(sc, 1 + pr)                                                         # :23


@pytest.mark.parametrize("patch", [3, 9])
def test_harris_matches_a_scipy_restatement(patch):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 56)).astype(np.uint8)
    ref = harris_numpy(img, patch, 0.08)
    # MATLAB memory is column-major: feed the transposed bytes ([v][u] = [col][row]) and transpose back
    got_cm = oracle.corner_response(np.ascontiguousarray(img.T), scorer=1, filter_size=patch, kappa=0.08).T
    got_rm = oracle.corner_response(img, scorer=1, filter_size=patch, kappa=0.08)
    assert ref.shape == img.shape
    np.testing.assert_allclose(got_cm, ref, rtol=1e-12, atol=1e-6)
    np.testing.assert_allclose(got_rm, ref, rtol=1e-12, atol=1e-6)   # the response is layout symmetric


def select_keypoints_numpy(scores, num, r):
    """old/selectKeypoints.m:5-14 on a [row, col] array; returns 1-based [row; col]."""
    t = np.pad(scores.astype(np.float64), r)
    out = np.zeros((2, num), np.int64)
    for i in range(num):
        k = int(np.argmax(t.T.ravel()))          # column-major first maximum (:8)
        col, row = divmod(k, t.shape[0])
        out[:, i] = [row + 1 - r, col + 1 - r]   # :9-11
        t[max(row - r, 0):row + r + 1, max(col - r, 0):col + r + 1] = 0
    return out


def test_greedy_selection_matches_numpy_restatement_with_ties():
    rng = np.random.default_rng(4)
    sc = rng.integers(0, 6, (24, 31)).astype(np.float64)  # many exact ties
    ref = select_keypoints_numpy(sc, 25, 2)
    # column-major memory: internal u = row, v = col
    ou, ov = oracle.select_keypoints_greedy(np.ascontiguousarray(sc.T), 25, 2)
    got = np.stack([ou + 1, ov + 1])
    np.testing.assert_array_equal(got, ref)


def test_mineigen_response_is_nonnegative_and_peaks_on_a_corner():
    img = np.zeros((64, 64), np.uint8); img[32:, 32:] = 200
    r = oracle.corner_response(img, scorer=0, filter_size=5)
    assert r.min() >= 0
    v, u = np.unravel_index(np.argmax(r), r.shape)
    assert abs(v - 32) <= 2 and abs(u - 32) <= 2
    assert r[10, 10] == 0 and r[32, 10] == 0  # flat area and a straight edge have a zero minor eigenvalue


def test_gauss_weights_sum_close_to_4096():
    for fs in (3, 5, 7, 9):
        w = oracle.gauss_weights(fs)
        assert abs(int(w.sum()) - 4096) <= fs and (w == w[::-1]).all()


def test_detection_order_ties_and_subpixel():
    resp = np.zeros((12, 12)); resp[5, 5] = 4.0; resp[5, 6] = 4.0; resp[8, 3] = 2.0
    lin, lu, lv, met = oracle.detect_corners(resp, 0.1)
    # the tie (5,5)-(5,6) is won by the lower linear index; ascending linear order
    assert list(lin) == [5 * 12 + 5, 8 * 12 + 3]
    assert lu[0] == pytest.approx(5.5) and lv[0] == 5.0   # parabola vertex between the equal samples
    assert lu[1] == 3.0 and lv[1] == 8.0


def test_distance_filter_is_strict():
    """modules/selectCandidateKeypoints.m:26: D < MinDistance, so d == MinDistance is KEPT."""
    img = np.zeros((40, 40), np.uint8); img[20:, 20:] = 255
    u, v, lin, met = oracle.select_candidates(img, [], [], min_distance=5.0, keep=1)
    assert len(u) == 1
    cu, cv = float(u[0]), float(v[0])
    kept = oracle.select_candidates(img, [cu + 3.0], [cv + 4.0], min_distance=5.0, keep=1)   # exactly 5 away
    dropped = oracle.select_candidates(img, [cu + 3.0], [cv + 3.9], min_distance=5.0, keep=1)
    assert len(kept[0]) == 1 and kept[2][0] == lin[0]
    assert len(dropped[0]) == 0 or dropped[2][0] != lin[0]


def test_select_uniform_spreads_and_caps():
    img, _, _ = synth.frame_pair(nv=120, nu=200, seed=11)
    u, v, lin, met = oracle.select_candidates(img, [], [], min_distance=0.0, min_quality=0.001, keep=40)
    assert len(u) == 40 and len(set(lin.tolist())) == 40
    assert (np.diff(lin) > 0).all()                      # ascending linear index
    # spread: every quadrant of the image holds some points
    q = (u > 100).astype(int) * 2 + (v > 60).astype(int)
    assert set(q.tolist()) == {0, 1, 2, 3}
    allu, _, _, _ = oracle.select_candidates(img, [], [], min_distance=0.0, min_quality=0.001, keep=100000)
    assert len(allu) < 100000                            # fewer corners than asked: all are returned


# ---------------------------------------------------------------- geometry
K = synth.KITTI_K


def test_camera_matrix_projects_like_the_textbook():
    R = synth.rot_y(7.0); C = np.array([0.4, -0.1, 1.5])
    cam = oracle.camera_matrix(synth.pose43(R, C), K)
    X = np.array([[1.0, 0.5, 12.0], [-3.0, 1.0, 30.0]])
    uvw = np.hstack([X, np.ones((2, 1))]) @ cam
    ref, depth = synth.project(K, R, C, X)
    np.testing.assert_allclose(uvw[:, :2] / uvw[:, 2:], ref, rtol=1e-12)
    np.testing.assert_allclose(uvw[:, 2], depth, rtol=1e-12)


def test_angle_known_answers():
    p0 = synth.pose43(np.eye(3), [1.0, 0, 0]); p1 = synth.pose43(np.eye(3), [0, 1.0, 0])
    assert oracle.angle_deg([0, 0, 0], p0, p1) == pytest.approx(90.0, abs=1e-12)
    assert oracle.angle_deg([0, 0, 0], p0, synth.pose43(np.eye(3), [2.0, 0, 0])) == pytest.approx(0.0, abs=1e-6)
    assert oracle.angle_deg([0, 0, 0], p0, synth.pose43(np.eye(3), [-1.0, 0, 0])) == pytest.approx(180.0, abs=1e-6)
    assert np.isnan(oracle.angle_deg([1.0, 0, 0], p0, p1))  # landmark at a camera centre: 0/0


def test_in_front_is_the_sign_of_the_third_column():
    cam = oracle.camera_matrix(synth.pose43(np.eye(3), [0, 0, 0]), K)
    assert list(oracle.in_front(cam, [[0, 0, 5.0], [0, 0, -5.0], [0, 0, 0.0]])) == [True, False, False]


def test_triangulation_recovers_noise_free_points():
    xy0, xy1, poses0, pose1, X = synth.triangulation_scene(200, noise=0.0)
    Xe, rep, ang, valid = oracle.triangulate(xy0, xy1, poses0, pose1, K)
    ok = ang > 0.2                                        # the zero-baseline poses (k = n_poses) are ill posed
    assert valid[ok].all()
    assert np.abs(Xe[ok] - X[ok]).max() / np.abs(X[ok]).max() < 1e-7
    assert rep[ok].max() < 1e-7


def test_triangulation_flags_points_behind_a_camera():
    pose0 = synth.pose43(np.eye(3), [0, 0, 0]); pose1 = synth.pose43(np.eye(3), [1.0, 0, 0])
    X = np.array([[0.5, 0.2, -8.0]])                     # behind both cameras
    xy0, _ = synth.project(K, np.eye(3), np.zeros(3), X); xy1, _ = synth.project(K, np.eye(3), np.array([1.0, 0, 0]), X)
    Xe, rep, ang, valid = oracle.triangulate(xy0, xy1, pose0[None], pose1, K)
    assert not valid[0] and np.allclose(Xe[0], X[0], atol=1e-8)


def test_triangulation_from_camera_matrices_is_the_same_triangulation():
    """the toolbox call's shape (modules/VisualOdometry.m:97-101): two 4 x 3 camera matrices instead of poses and K"""
    xy0, xy1, poses0, pose1, X = synth.triangulation_scene(64, noise=0.3)
    cam1 = oracle.camera_matrix(pose1, K)
    Xe, rep, ang, valid = oracle.triangulate(xy0, xy1, poses0, pose1, K)
    for i in range(len(xy0)):                            # one candidate a call, as the class method's loop makes it
        Xi, ri, vi = oracle.triangulate_cam(xy0[i], xy1[i], oracle.camera_matrix(poses0[i], K), cam1)
        assert Xi.tobytes() == Xe[i:i + 1].tobytes() and ri[0] == rep[i] and vi[0] == valid[i]
    # n pairs under one pair of matrices, noise free: the points come back
    pose0 = synth.pose43(synth.rot_y(2.0), [0.0, 0.0, 0.0]); pose2 = synth.pose43(synth.rot_y(-1.0), [1.5, 0.1, 0.2])
    Xt = np.random.default_rng(3).uniform([-10, -3, 8], [10, 3, 60], (100, 3))
    a, _ = synth.project(K, pose0[:3], pose0[3], Xt); b, _ = synth.project(K, pose2[:3], pose2[3], Xt)
    Xr, rr, vr = oracle.triangulate_cam(a, b, oracle.camera_matrix(pose0, K), oracle.camera_matrix(pose2, K))
    assert vr.all() and np.abs(Xr - Xt).max() < 1e-7 and rr.max() < 1e-8


def test_triangulation_against_a_numpy_svd_restatement():
    """Second, independent restatement of the two-view DLT (modules/VisualOdometry.m:93-101, old/linearTriangulation.m:28-39)
    on NOISY observations: camera matrices by plain matrix products, the null vector by numpy's SVD (LAPACK) instead of
    the oracle's one-sided Jacobi, error and validity from the projections.  The oracle's Jacobi order and stopping rule
    were redefined together with the kernel once (round 1); this holds the result, not the path to it."""
    xy0, xy1, poses0, pose1, _ = synth.triangulation_scene(400, noise=0.5)
    Xe, rep, ang, valid = oracle.triangulate(xy0, xy1, poses0, pose1, K)

    def cam(pose):                                       # [Rext; text] * K^T with Rext = R^T, text = -t R^T
        R, t = pose[:3], pose[3]
        return np.vstack([R.T, -t @ R.T]) @ K.T

    c1 = cam(pose1)
    worst = 0.0
    for i in range(len(xy0)):
        c0 = cam(poses0[i].reshape(4, 3))
        P0, P1 = c0.T, c1.T                              # 3 x 4
        A = np.stack([xy0[i, 0] * P0[2] - P0[0], xy0[i, 1] * P0[2] - P0[1], xy1[i, 0] * P1[2] - P1[0], xy1[i, 1] * P1[2] - P1[1]])
        s = np.linalg.svd(A)[1]
        h = np.linalg.svd(A)[2][-1]
        X = h[:3] / h[3]
        e = []
        w = []
        for c, xy in ((c0, xy0[i]), (c1, xy1[i])):
            p = np.append(X, 1.0) @ c
            w.append(p[2]); e.append(np.hypot(p[0] / p[2] - xy[0], p[1] / p[2] - xy[1]))
        if s[2] / s[0] < 1e-9:                           # (near-)zero baseline: the null space is not a line
            continue
        worst = max(worst, np.abs(Xe[i] - X).max() / max(1.0, np.abs(X).max()))
        assert abs(rep[i] - 0.5 * (e[0] + e[1])) < 1e-6 * max(1.0, rep[i])
        if min(abs(w[0]), abs(w[1])) > 1e-6:
            assert valid[i] == (w[0] > 0 and w[1] > 0)
        d0, d1 = X - poses0[i].reshape(4, 3)[3], X - pose1[3]
        cosang = np.clip(d0 @ d1 / (np.linalg.norm(d0) * np.linalg.norm(d1)), -1, 1)
        assert abs(ang[i] - np.degrees(np.arccos(cosang))) < 1e-6
    assert worst < 1e-9, worst                           # 2e-12 observed


def test_reprojection_error_is_the_pixel_distance():
    R = synth.rot_y(3.0); C = np.array([0.1, 0.0, 0.5])
    X = np.array([[2.0, 1.0, 20.0], [-4.0, 0.5, 35.0]])
    xy, _ = synth.project(K, R, C, X)
    err = oracle.reproj_error(xy + np.array([[3.0, 4.0], [0.0, 0.0]]), X, synth.pose43(R, C), K)
    np.testing.assert_allclose(err, [5.0, 0.0], atol=1e-9)


# ---------------------------------------------------------------- P3P / RANSAC (N9)
def test_quartic_known_roots():
    roots = oracle.solve_quartic(np.poly([1.0, 2.0, 3.0, 4.0]))      # (x-1)(x-2)(x-3)(x-4)
    assert np.abs(roots.imag).max() < 1e-9
    np.testing.assert_allclose(np.sort(roots.real), [1, 2, 3, 4], atol=1e-9)
    r2 = oracle.solve_quartic(np.poly([1j, -1j, 1 + 2j, 1 - 2j]).real)  # complex pairs +-i, 1+-2i
    got = sorted(r2, key=lambda z: (round(z.real, 6), round(z.imag, 6)))
    np.testing.assert_allclose(got, [-1j, 1j, 1 - 2j, 1 + 2j], atol=1e-9)
    # (a biquadratic, beta = 0, makes w = 0 in old/solveQuartic.m:59-64 -- the reference's own
    #  formula degenerates there and so does this restatement; P3P never produces that case)


def random_pose(rng):
    A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    return Q, rng.normal(size=3)


def test_p3p_reproduces_the_pose():
    """Kneip's construction: one of the four solutions is the true pose."""
    rng = np.random.default_rng(8)
    for _ in range(50):
        R_wc, C = random_pose(rng)                        # cam->world rotation and centre
        Pc = rng.uniform(-2, 2, (3, 3)) + np.array([0, 0, 6.0])
        Pw = Pc @ R_wc.T + C
        f = Pc / np.linalg.norm(Pc, axis=1, keepdims=True)
        poses = oracle.p3p(Pw, f)
        errs = [max(np.abs(p[:3] - C).max(), np.abs(p[3:].reshape(3, 3) - R_wc).max()) for p in poses]
        assert min(errs) < 1e-7


def test_p3p_collinear_points_give_zeros():
    P = np.array([[0, 0, 5.0], [1, 1, 6.0], [2, 2, 7.0]])
    f = P / np.linalg.norm(P, axis=1, keepdims=True)
    assert not oracle.p3p(P, f).any()                     # old/p3p.m:56-58


def test_philox_samples_are_distinct_in_range_and_seeded():
    seen = set()
    for h in range(300):
        i = oracle.philox_sample3(4095, h, 7)
        assert len(set(i.tolist())) == 3 and i.min() >= 0 and i.max() < 7
        seen.add(tuple(i))
    assert len(seen) > 100
    assert (oracle.philox_sample3(1, 5, 1000) == oracle.philox_sample3(1, 5, 1000)).all()
    assert (oracle.philox_sample3(1, 5, 1000) != oracle.philox_sample3(2, 5, 1000)).any()


def test_ransac_finds_the_inliers():
    """20 % gross outliers, fixed hypothesis stream."""
    xy, X, R_cw, C, is_out = synth.pnp_scene(400, noise=0.3)
    r = oracle.p3p_ransac(xy, X, K, num_hyp=300, max_reproj=2.0, seed=4095)
    assert r["status"] == 0
    assert (r["inliers"] & is_out).sum() <= 2 and (r["inliers"] & ~is_out).sum() > 0.9 * (~is_out).sum()
    np.testing.assert_allclose(r["R"], R_cw, atol=5e-3)
    np.testing.assert_allclose(-r["R"].T @ r["t"], C, atol=0.15)
    msac = oracle.p3p_ransac(xy, X, K, num_hyp=300, max_reproj=2.0, seed=4095, mode=1, min_inliers=4)
    assert msac["status"] == 0 and (msac["inliers"] & ~is_out).sum() > 0.9 * (~is_out).sum()


def test_ransac_statuses_and_adaptive_stop():
    xy, X, *_ = synth.pnp_scene(100, noise=0.1, outlier_frac=0.0)
    assert oracle.p3p_ransac(xy[:3], X[:3], K, num_hyp=10)["status"] == 1
    junk = oracle.p3p_ransac(np.random.default_rng(0).uniform(0, 300, (50, 2)), X[:50], K, num_hyp=20, min_inliers=30)
    assert junk["status"] == 2 and not junk["inliers"].any()
    full = oracle.p3p_ransac(xy, X, K, num_hyp=500, confidence=0.0)
    early = oracle.p3p_ransac(xy, X, K, num_hyp=500, confidence=0.99)
    assert full["trials"] == 500 and early["trials"] < 20      # all-inlier data stops after a few hypotheses
    table = np.array([oracle.philox_sample3(77, h, 100) for h in range(500)])
    same = oracle.p3p_ransac(xy, X, K, num_hyp=500, seed=77)
    tab = oracle.p3p_ransac(xy, X, K, num_hyp=500, seed=1, sample_idx=table)
    assert same["best_hyp"] == tab["best_hyp"] and (same["inliers"] == tab["inliers"]).all()


def test_frame_step_restatement_runs_a_sequence():
    """oracle/vo_frame.py (modules/VisualOdometry.m:264-346): landmarks are promoted, the cap holds,
    candidates age, and the pose follows the camera on a 3D-consistent synthetic sequence."""
    from oracle import vo_frame
    K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
    frames, poses, lift = synth.plane_sequence(6, K)
    u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=120)
    st = vo_frame.new_state()
    st.keypoints = np.stack([u, v], 1).astype(np.float64)
    st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1]); st.reproError = np.zeros(len(u))
    promoted = 0
    for k in range(1, 6):
        st, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, K, k, maxNumLandmarks=150, threads=8)
        assert status == 0 and len(st.landmarks) <= 150
        assert len(st.candidate_keypoints) == len(st.candidate_first_poses) == len(st.candidate_time_indxs)
        assert np.abs(pose[3] - poses[k][3]).max() < 0.3 and np.abs(pose[:3] - poses[k][:3]).max() < 0.05
        promoted += info["promoted"]
    assert promoted > 0 and st.candidate_time_indxs.min() < -1 and (st.reproError < 3.0).all()


def test_eight_point_and_essential_msac():
    """N10 (oracle/essential.c): exact correspondences give the true E up to sign; the MSAC separates
    gross outliers; samples are 8 distinct indices."""
    rng = np.random.default_rng(2)
    n = 600
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-5, 5, n), rng.uniform(8, 60, n)], 1)
    R, C = synth.rot_y(3.0), np.array([1.0, -0.1, 0.2])
    K = synth.KITTI_K
    p0, _ = synth.project(K, np.eye(3), np.zeros(3), X)
    p1, _ = synth.project(K, R, C, X)
    t = -R @ C
    Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    Et *= np.sqrt(2) / np.linalg.norm(Et)
    Ki = np.linalg.inv(K)
    x0 = (np.c_[p0[:8], np.ones(8)] @ Ki.T)[:, :2]; x1 = (np.c_[p1[:8], np.ones(8)] @ Ki.T)[:, :2]
    ok, E = oracle.eight_point(x0, x1)
    assert ok and min(np.abs(E - Et).max(), np.abs(E + Et).max()) < 1e-8
    np.testing.assert_allclose(np.linalg.svd(E)[1], [1, 1, 0], atol=1e-12)
    out = rng.random(n) < 0.25
    p1n = p1 + rng.normal(0, 0.03, p1.shape)
    p1n[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
    r = oracle.essential_ransac(p0, p1n, K, num_hyp=500, threads=8)
    assert r["status"] == 0 and (r["inliers"] & out).sum() <= 2 and r["inliers"][~out].mean() > 0.6
    assert oracle.essential_ransac(p0[:7], p1[:7], K)["status"] == 1
    for h in range(50):
        s = oracle.philox_sample8(9, h, 11)
        assert len(set(s)) == 8 and s.min() >= 0 and s.max() < 11


def test_bootstrap_restatement_on_a_relief_scene():
    from oracle import bootstrap as ob
    K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
    frames, poses, lift = synth.relief_sequence(3, K)
    kp, X, rep, pose, info = ob.bootstrap(frames[0], frames[2], K, threads=8)
    Ct = poses[2][3]
    assert len(kp) == len(X) == len(rep) > 500 and info["attempts"] == 1
    assert np.abs(pose[:3] - poses[2][:3]).max() < 0.02 and np.abs(pose[3] - Ct / np.linalg.norm(Ct)).max() < 0.15
    assert np.median(rep) < 0.2 and (X[:, 2] > 0).all()
