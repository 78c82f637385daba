"""Parity of the HIP path (through the C ABI of libvo_hip.so) against the oracle and the committed
golden vectors.  Bars (BASELINE.json north_star): selection indices and inlier sets under identical
hypotheses bit-exact; tracked positions within 0.01 px; landmarks and poses within 1e-4 relative
(here: far tighter, both sides being IEEE double in one operation order)."""
import numpy as np
import pytest

from oracle import oracle
from visual_odometry_amd import modules, ops, synth
from tests.test_golden import KLT_CFG, SEL_CFG

pytestmark = pytest.mark.gpu
K = synth.KITTI_K
PX_TOL = 0.01  # pixels, north_star
FB_TOL = 1e-3  # pixels: validity may differ from the oracle's only where the forward-backward error sits this close to
               # MaxBidirectionalError (measured on every scenario below: no disagreement at all, scripts/klt_disagree.py)


def assert_validity(valid, ovalid, ofb, max_bidir, min_both):
    """Validity against the oracle's: equal, except for points whose forward-backward error (the oracle's) lies
    within FB_TOL of the threshold -- the only place where fp32 against fp64 arithmetic may legitimately flip it.
    Returns the mask of points valid on both sides, which must hold at least `min_both` points."""
    dis = valid != ovalid
    if dis.any():
        assert max_bidir >= 0 and (np.abs(ofb[dis] - max_bidir) < FB_TOL).all(), \
            f"{dis.sum()} validity flips away from the threshold: fb = {ofb[dis][:8]}"
    both = valid & ovalid
    assert both.sum() >= min_both, f"only {both.sum()} points are valid on both sides"
    return both


def klt_gpu(img1, img2, pu, pv, levels, wu, wv, max_iters, max_bidir, col_major=False):
    """internal (u,v) zero-based -> ops.klt_track in either memory layout."""
    if col_major:  # MATLAB layout: the same bytes seen as H x W column-major, u = row (y), v = column (x)
        a, b = np.asfortranarray(img1.T), np.asfortranarray(img2.T)
        pts = np.stack([pv, pu], 1)
        out, valid, score = ops.klt_track(a, b, pts, levels, max_bidir, (wu, wv), max_iters, index_base=0)
        return out[:, 1], out[:, 0], valid, score
    out, valid, score = ops.klt_track(img1, img2, np.stack([pu, pv], 1), levels, max_bidir, (wv, wu), max_iters,
                                      index_base=0)
    return out[:, 0], out[:, 1], valid, score


@pytest.mark.parametrize("tag", ["s", "l", "m"])
@pytest.mark.parametrize("col_major", [False, True])
def test_klt_matches_golden(ctx, golden, tag, col_major):
    g, kw = golden, KLT_CFG[tag]
    qu, qv, valid, score = klt_gpu(g["klt_img1"], g["klt_img2"], g["klt_pu"], g["klt_pv"], kw["levels"], kw["wu"],
                                   kw["wv"], kw["max_iters"], kw["max_bidir"], col_major)
    both = assert_validity(valid, g[f"klt_{tag}_valid"], g[f"klt_{tag}_fb"], kw["max_bidir"], 40)
    d = np.hypot(qu - g[f"klt_{tag}_qu"], qv - g[f"klt_{tag}_qv"])
    assert d[both].max() < PX_TOL
    assert np.abs(score - g[f"klt_{tag}_score"])[both].max() < 1e-3


def test_klt_typed_empties_and_dtype_following(ctx):
    img = np.zeros((64, 64), np.uint8)
    p, v, s = modules.KLTTracker().track(img, img, np.zeros((0, 2), np.float32))
    assert p.shape == (0, 2) and p.dtype == np.float64 and v.shape == (0,) and s.shape == (0,)  # KLTTracker.m:33-36
    img1, img2, _ = synth.frame_pair(nv=120, nu=160, seed=3)
    pts = np.array([[60.0, 50.0], [100.5, 70.25]])
    p64, v64, s64 = modules.KLTTracker().track(img1, img2, pts)
    p32, v32, s32 = modules.KLTTracker().track(img1, img2, pts.astype(np.float32))
    assert p64.dtype == np.float64 and p32.dtype == np.float32 and s32.dtype == np.float32
    assert np.abs(p64 - p32).max() < 1e-3 and (v64 == v32).all()


def test_klt_matlab_one_based_convention(ctx):
    """[x y] 1-based on a column-major image == the internal zero-based call on the same bytes."""
    img1, img2, truth = synth.frame_pair(nv=120, nu=160, seed=4, shift=(2.2, -1.4))
    pu, pv = synth.keypoints(50, nv=120, nu=160, seed=5, border=22)
    trk = modules.KLTTracker(NumPyramidLevels=3, MaxBidirectionalError=1, BlockSize=(31, 31), MaxIterations=30)
    p, v, s = trk.track(np.asfortranarray(img1), np.asfortranarray(img2), np.stack([pu + 1, pv + 1], 1))
    tu, tv = truth(pu, pv)
    assert v.all() and np.hypot(p[:, 0] - 1 - tu, p[:, 1] - 1 - tv).max() < 0.06
    qu, qv, valid, *_ = oracle.klt_track(img1, img2, pu, pv, levels=3, max_bidir=1.0, wu=31, wv=31, max_iters=30)
    assert np.hypot(p[:, 0] - 1 - qu, p[:, 1] - 1 - qv).max() < PX_TOL


def test_klt_bad_arguments_raise(ctx):
    img = np.zeros((64, 64), np.uint8)
    with pytest.raises(ValueError):
        modules.KLTTracker(BlockSize=(40, 41))
    with pytest.raises(Exception):
        ops.klt_track(img, img, np.ones((3, 2)), block=(101, 101))
    with pytest.raises(ValueError):
        ops.klt_track(img, np.zeros((32, 64), np.uint8), np.ones((3, 2)))


def test_klt_kitti_size_against_oracle_and_truth(ctx):
    """BASELINE stress shape: 1241 x 376, 4 levels, 41 x 41, 50 iterations; 50k points tracked on
    the GPU, the first 1500 also by the oracle."""
    img1, img2, truth = synth.frame_pair(shift=(3.3, -2.6), zoom=0.004, noise=1.0)
    pu, pv = synth.keypoints(50000)
    qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, 4, 41, 41, 50, 2.0)
    tu, tv = truth(pu, pv)
    err = np.hypot(qu - tu, qv - tv)
    assert valid.mean() > 0.99 and np.median(err[valid]) < 0.03 and np.percentile(err[valid], 99) < 0.1
    n = 1500
    ou, ov, ovalid, oscore, ofb, _ = oracle.klt_track(img1, img2, pu[:n], pv[:n], threads=8)
    both = assert_validity(valid[:n], ovalid, ofb, 2.0, 1480)
    assert np.hypot(qu[:n] - ou, qv[:n] - ov)[both].max() < PX_TOL
    # idempotence of the frame cache: a second call on the same pair gives identical bits
    qu2, qv2, valid2, _ = klt_gpu(img1, img2, pu, pv, 4, 41, 41, 50, 2.0)
    assert np.array_equal(qu, qu2) and np.array_equal(qv, qv2) and np.array_equal(valid, valid2)


@pytest.mark.parametrize("wu,wv,levels", [(7, 7, 3), (21, 21, 3), (23, 19, 2), (35, 31, 4), (41, 41, 4), (45, 41, 3),
                                          (41, 45, 3), (51, 51, 4), (55, 55, 2), (3, 55, 2)])
def test_klt_window_classes_and_borders(ctx, wu, wv, levels):
    """Every window class of k_lk (CfgS/M/L/X, square and not), with keypoints right up to the image
    border: both staging forms (16-byte rows with per-lane row clamps, 4-byte rows with column
    clamps) and the lost / flat exits, against the oracle."""
    nv, nu = 150, 210
    img1, img2, _ = synth.frame_pair(nv=nv, nu=nu, seed=11, shift=(1.7, -1.2), zoom=0.003)
    pu, pv = synth.keypoints(400, nv=nv, nu=nu, seed=12, border=0.0)
    qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, levels, wu, wv, 30, 1.5)
    ou, ov, ovalid, oscore, ofb, _ = oracle.klt_track(img1, img2, pu, pv, levels=levels, max_bidir=1.5, wu=wu, wv=wv,
                                                      max_iters=30, threads=8)
    both = assert_validity(valid, ovalid, ofb, 1.5, 370)
    assert (~ovalid).sum() >= 5                           # points that leave the image or fail the check are in the set
    assert np.hypot(qu - ou, qv - ov)[both].max() < PX_TOL
    assert np.abs(score - oscore)[both].max() < 1e-3


@pytest.mark.parametrize("nv,nu", [(40, 48), (33, 200), (200, 37), (24, 24)])
def test_klt_images_smaller_than_the_patch(ctx, nv, nu):
    """Images narrower / lower than the staged patch: every level is replicate-clamped."""
    img1, img2, _ = synth.frame_pair(nv=nv, nu=nu, seed=13, shift=(0.8, 0.6), margin=16)
    pu, pv = synth.keypoints(64, nv=nv, nu=nu, seed=14, border=2.0)
    qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, 3, 31, 31, 30, 2.0)
    ou, ov, ovalid, oscore, ofb, _ = oracle.klt_track(img1, img2, pu, pv, levels=3, max_bidir=2.0, wu=31, wv=31, max_iters=30)
    both = assert_validity(valid, ovalid, ofb, 2.0, 60)
    assert np.hypot(qu - ou, qv - ov)[both].max() < PX_TOL
    assert np.abs(score - oscore)[both].max() < 1e-3


def test_klt_points_outside_the_image_or_not_finite(ctx):
    """A caller may hand over anything: NaN, infinities, positions outside the image, positions exactly on
    the border.  Such points come back invalid (as the oracle has them: N4's lost test / flat exit), nothing
    faults, and the good points beside them get the bits they get without them."""
    nv, nu = 150, 210
    img1, img2, _ = synth.frame_pair(nv=nv, nu=nu, seed=15, shift=(1.2, -0.7))
    gu, gv = synth.keypoints(60, nv=nv, nu=nu, seed=16, border=25.0)
    odd = [np.nan, np.inf, -np.inf, -0.5, -30.0, nu - 0.75, nu + 40.0, 1e12, -1e12, 3e38, 0.0, float(nu - 1)]
    bu = np.array(odd + [50.0] * len(odd))
    bv = np.array([40.0] * len(odd) + [np.nan, np.inf, -np.inf, -0.5, -30.0, nv - 0.75, nv + 40.0, 1e12, -1e12, 3e38, 0.0,
                                       float(nv - 1)])
    pu = np.concatenate([gu[:30], bu, gu[30:]]); pv = np.concatenate([gv[:30], bv, gv[30:]])
    bad = np.zeros(len(pu), bool); bad[30:30 + len(bu)] = True
    for wu, wv, levels in ((31, 31, 3), (41, 41, 4), (7, 7, 2)):
        for cm in (False, True):
            qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, levels, wu, wv, 30, 1.5, cm)
            ou, ov, ovalid, oscore, *_ = oracle.klt_track(img1, img2, pu, pv, levels=levels, max_bidir=1.5, wu=wu, wv=wv,
                                                          max_iters=30)
            outside = bad & ~((pu >= 0) & (pu <= nu - 1) & (pv >= 0) & (pv <= nv - 1))
            assert outside.sum() == 20 and not valid[outside].any() and not ovalid[outside].any()
            assert np.array_equal(valid, ovalid)
            both = valid & ovalid
            assert both[~bad].sum() > 40 and np.hypot(qu[both] - ou[both], qv[both] - ov[both]).max() < PX_TOL
            cu, cv, cvalid, cscore = klt_gpu(img1, img2, gu, gv, levels, wu, wv, 30, 1.5, cm)
            assert np.array_equal(qu[~bad], cu) and np.array_equal(qv[~bad], cv) and np.array_equal(valid[~bad], cvalid)
            assert np.array_equal(score[~bad], cscore)


@pytest.mark.parametrize("nv,nu", [(8, 8), (8, 60), (90, 8), (9, 11)])
def test_klt_smallest_images(ctx, nv, nu):
    """The smallest shapes the entry point takes (8 x 8, top level at least 2 x 2); beyond that it is an error."""
    img1, img2, _ = synth.frame_pair(nv=nv, nu=nu, seed=19, shift=(0.4, 0.3), margin=8)
    pu, pv = synth.keypoints(40, nv=nv, nu=nu, seed=20, border=0.0)
    qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, 3, 7, 7, 30, 2.0)
    ou, ov, ovalid, oscore, ofb, _ = oracle.klt_track(img1, img2, pu, pv, levels=3, max_bidir=2.0, wu=7, wv=7, max_iters=30)
    both = assert_validity(valid, ovalid, ofb, 2.0, 20)
    assert (~ovalid).sum() >= 5                           # and points the oracle rejects, rejected here too
    assert np.hypot(qu[both] - ou[both], qv[both] - ov[both]).max() < PX_TOL
    with pytest.raises(Exception):
        ops.klt_track(img1[:7], img2[:7], np.ones((3, 2)))
    if min(nv, nu) == 8:
        with pytest.raises(Exception):                      # 8 -> 4 -> 2 -> 1: the fourth level is below 2 x 2
            ops.klt_track(img1, img2, np.ones((3, 2)), 4)


def test_selection_existing_points_outside_the_image_or_not_finite(ctx):
    """Existing keypoints of any value: the distance filter's grid clamps them into its border cells (NaN into
    cell 0); the verdict stays the oracle's all-pairs one, bit for bit."""
    img, _, _ = synth.frame_pair(nv=150, nu=210, seed=17, noise=1.0)
    eu, ev = synth.keypoints(300, nv=150, nu=210, seed=18, border=0.0)
    odd = np.array([np.nan, np.inf, -np.inf, -3.0, -1e9, 212.5, 1e9, 3e38, 0.0, 209.0], np.float32)
    eu = np.concatenate([eu.astype(np.float32), odd, np.full(len(odd), 100, np.float32)])
    ev = np.concatenate([ev.astype(np.float32), np.full(len(odd), 70, np.float32), odd])
    for cm in (False, True):
        u, v, lin, met = sel_gpu(img, eu, ev, cm, min_distance=7.0, min_quality=0.001, filter_size=5, keep=60)
        ru, rv, rlin, rmet = oracle.select_candidates(img, eu, ev, 7.0, 0.001, 5, 60, col_major=cm)
        np.testing.assert_array_equal(lin, rlin)
        np.testing.assert_array_equal(u, ru); np.testing.assert_array_equal(v, rv)
        assert len(lin) > 10


@pytest.mark.parametrize("nv,nu,n", [(600, 800, 2000), (768, 1024, 5000)])
def test_klt_malaga_shapes(ctx, nv, nu, n):
    """BASELINE configs on the Malaga shapes (utils/dataLoaderMalaga.m:54-55 loads 800 x 600; BASELINE
    quotes 1024 x 768): all n points on the GPU, the first 300 also by the oracle."""
    img1, img2, truth = synth.frame_pair(nv=nv, nu=nu, seed=21, shift=(-4.1, 2.7), zoom=0.003, noise=1.0)
    pu, pv = synth.keypoints(n, nv=nv, nu=nu, seed=22)
    qu, qv, valid, score = klt_gpu(img1, img2, pu, pv, 4, 41, 41, 50, 2.0)
    tu, tv = truth(pu, pv)
    err = np.hypot(qu - tu, qv - tv)
    assert valid.mean() > 0.99 and np.percentile(err[valid], 99) < 0.1
    m = 300
    ou, ov, ovalid, _, ofb, _ = oracle.klt_track(img1, img2, pu[:m], pv[:m], threads=8)
    both = assert_validity(valid[:m], ovalid, ofb, 2.0, 295)
    assert np.hypot(qu[:m] - ou, qv[:m] - ov)[both].max() < PX_TOL


# ---------------------------------------------------------------- selection
def sel_gpu(img, eu, ev, col_major, **kw):
    """internal-coordinate call in either layout; returns u, v, lin, metric like the oracle."""
    if col_major:
        xy, lin, met = ops.select_candidates(np.asfortranarray(img.T), np.stack([ev, eu], 1), index_base=0,
                                             want_all=True, **kw)
        return xy[:, 1], xy[:, 0], lin, met
    xy, lin, met = ops.select_candidates(img, np.stack([eu, ev], 1), index_base=0, want_all=True, **kw)
    return xy[:, 0], xy[:, 1], lin, met


@pytest.mark.parametrize("tag", ["me_rm", "me_cm", "ha_rm"])
def test_selection_matches_golden_bit_exactly(ctx, golden, tag):
    g, c = golden, SEL_CFG[tag]
    u, v, lin, met = sel_gpu(g["sel_img"], g["sel_eu"], g["sel_ev"], c["col_major"], min_distance=7.0,
                             min_quality=0.001, filter_size=c["filter_size"], keep=25,
                             scorer="harris" if c["scorer"] else "mineigen", kappa=c.get("kappa", 0.08))
    np.testing.assert_array_equal(lin, g[f"sel_{tag}_lin"])
    np.testing.assert_array_equal(u, g[f"sel_{tag}_u"])
    np.testing.assert_array_equal(v, g[f"sel_{tag}_v"])
    np.testing.assert_array_equal(met, g[f"sel_{tag}_metric"])


def test_responses_match_golden_bit_exactly(ctx, golden):
    g = golden
    np.testing.assert_array_equal(ops.corner_response(g["harris_img"], "harris", 9, 0.08), g["harris_resp"])
    np.testing.assert_array_equal(ops.corner_response(g["harris_img"], "mineigen", 5), g["mineig_resp"])
    rc = ops.select_keypoints_greedy(g["harris_resp"], 10, 3, index_base=0)
    np.testing.assert_array_equal(rc[1], g["greedy_u"]); np.testing.assert_array_equal(rc[0], g["greedy_v"])


@pytest.mark.parametrize("scorer,fs", [("mineigen", 5), ("mineigen", 9), ("harris", 3), ("harris", 9), ("mineigen", 15)])
def test_response_kitti_size_bit_exact(ctx, scorer, fs):
    img, _, _ = synth.frame_pair(seed=41)
    got = ops.corner_response(img, scorer, fs, 0.08)
    ref = oracle.corner_response(img, 0 if scorer == "mineigen" else 1, fs, 0.08)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("keep,n_exist", [(100, 1500), (5000, 20000), (100, 0), (100, 50000), (5000, 50000)])
def test_selection_kitti_size_bit_exact(ctx, keep, n_exist):
    """Live options of modules/VisualOdometry.m:330-335 and the stress variant of SURVEY 8d; n_exist = 50000 is the
    size bench.py's step runs the filter at (device.FrameWorkload: every tracked point is an existing point -- 50k
    discs of radius 7 cover a KITTI frame many times over, so few corners or none survive; the oracle says which)."""
    img, _, _ = synth.frame_pair(seed=42, noise=1.0)
    eu, ev = synth.keypoints(n_exist, seed=43, border=0.0) if n_exist else (np.zeros(0), np.zeros(0))
    eu = eu.astype(np.float32); ev = ev.astype(np.float32)
    for col_major in (False, True):
        u, v, lin, met = sel_gpu(img, eu, ev, col_major, min_distance=7.0, min_quality=0.001, filter_size=5, keep=keep)
        ru, rv, rlin, rmet = oracle.select_candidates(img, eu, ev, 7.0, 0.001, 5, keep, col_major=col_major)
        np.testing.assert_array_equal(lin, rlin)
        np.testing.assert_array_equal(u, ru); np.testing.assert_array_equal(v, rv)
        np.testing.assert_array_equal(met, rmet)
        assert len(lin) > 0 or n_exist >= 50000


def test_selection_ties_and_strict_distance(ctx):
    img = np.zeros((40, 40), np.uint8); img[20:, 20:] = 255; img[:10, :10] = 255  # symmetric corners: equal metrics
    for keep in (1, 2, 8):
        u, v, lin, met = sel_gpu(img, np.zeros(0), np.zeros(0), False, min_distance=5.0, keep=keep)
        ru, rv, rlin, rmet = oracle.select_candidates(img, [], [], min_distance=5.0, keep=keep)
        np.testing.assert_array_equal(lin, rlin)
    cu, cv = float(ru[0]), float(rv[0])
    kept = sel_gpu(img, np.array([cu + 3.0], np.float32), np.array([cv + 4.0], np.float32), False, min_distance=5.0, keep=50)
    drop = sel_gpu(img, np.array([cu + 3.0], np.float32), np.array([cv + 3.9], np.float32), False, min_distance=5.0, keep=50)
    assert rlin[0] in kept[2] and rlin[0] not in drop[2]   # d == MinDistance is kept (strict <, :26)


def test_greedy_selection_with_ties_bit_exact(ctx):
    rng = np.random.default_rng(4)
    sc = rng.integers(0, 6, (24, 31)).astype(np.float64)
    ou, ov = oracle.select_keypoints_greedy(np.ascontiguousarray(sc.T), 25, 2)
    rc = modules.selectKeypoints(np.asfortranarray(sc), 25, 2)           # MATLAB layout, 1-based [row; col]
    np.testing.assert_array_equal(rc[0], ou + 1); np.testing.assert_array_equal(rc[1], ov + 1)
    zero = modules.selectKeypoints(np.zeros((8, 8), order="F"), 2, 2)    # all-zero scores: padded corner wins
    assert (zero == -1).all()


# ---------------------------------------------------------------- geometry
def test_triangulation_matches_golden(ctx, golden):
    g = golden
    for cm in (False, True):
        X, rep, ang, valid = ops.triangulate_candidates(g["tri_xy0"], g["tri_xy1"], g["tri_poses0"], g["tri_pose1"], K,
                                                        col_major=cm)
        np.testing.assert_array_equal(valid, g["tri_valid"])
        np.testing.assert_allclose(X, g["tri_X"], rtol=1e-12)
        np.testing.assert_allclose(rep, g["tri_reproj"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ang, g["tri_angle"], rtol=1e-9)


def test_triangulation_pose_table_and_scale(ctx):
    xy0, xy1, poses0, pose1, Xt = synth.triangulation_scene(50000)
    uniq, idx = np.unique(poses0.reshape(len(poses0), -1), axis=0, return_inverse=True)
    X, rep, ang, valid = ops.triangulate_candidates(xy0, xy1, uniq.reshape(-1, 4, 3), pose1, K, first_pose_idx=idx + 1)
    Xo, repo, ango, valido = oracle.triangulate(xy0, xy1, poses0, pose1, K)
    np.testing.assert_array_equal(valid, valido)
    np.testing.assert_allclose(X, Xo, rtol=1e-12); np.testing.assert_allclose(rep, repo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ang, ango, rtol=1e-9)
    with pytest.raises(Exception):
        ops.triangulate_candidates(xy0[:4], xy1[:4], uniq.reshape(-1, 4, 3)[:2], pose1, K)  # P != C without a table
    bad = ops.triangulate_candidates(xy0[:4], xy1[:4], uniq.reshape(-1, 4, 3), pose1, K, first_pose_idx=[1, 2, 99, 0])
    assert not bad[3][2:].any()                                    # indices outside the table (99, 0 in 1-based) are flagged invalid
    _, _, _, ok2 = oracle.triangulate(xy0[:2], xy1[:2], uniq.reshape(-1, 4, 3)[:2], pose1, K)
    np.testing.assert_array_equal(bad[3][:2], ok2)                 # the good ones beside them are what the oracle says


def test_triangulate_with_camera_matrices(ctx):
    """vo_triangulate_points -- the toolbox call `triangulate(p1, p2, camMatrix1, camMatrix2)` as the unmodified class
    method makes it (modules/VisualOdometry.m:97-101), one candidate a call, and batched -- against the oracle, and
    against the batched candidate entry point, which must agree bit for bit (same DLT on the same matrices)."""
    xy0, xy1, poses0, pose1, Xt = synth.triangulation_scene(3000)
    cam1 = oracle.camera_matrix(pose1, K)
    Xc, repc, angc, validc = ops.triangulate_candidates(xy0, xy1, poses0.reshape(-1, 4, 3), pose1, K)
    for i in (0, 1, 17, 2999):                            # one candidate a call
        cam0 = oracle.camera_matrix(poses0[i], K)
        for cm in (False, True):
            X, rep, valid = ops.triangulate(xy0[i], xy1[i], cam0, cam1, col_major=cm)
            Xo, repo, valido = oracle.triangulate_cam(xy0[i], xy1[i], cam0, cam1)
            assert valid[0] == valido[0]
            np.testing.assert_allclose(X, Xo, rtol=1e-12); np.testing.assert_allclose(rep, repo, rtol=1e-9, atol=1e-12)
            # the device builds the camera matrix of the batched entry point with the oracle's expression
            assert X.tobytes() == np.ascontiguousarray(Xc[i:i + 1]).tobytes() and rep[0] == repc[i] and valid[0] == validc[i]
    sel = np.flatnonzero((poses0.reshape(len(poses0), -1) == poses0.reshape(len(poses0), -1)[0]).all(1))   # one first pose
    assert len(sel) > 100
    cam0 = oracle.camera_matrix(poses0[sel[0]], K)
    for cm in (False, True):
        X, rep, valid = ops.triangulate(xy0[sel], xy1[sel], cam0, cam1, col_major=cm)
        Xo, repo, valido = oracle.triangulate_cam(xy0[sel], xy1[sel], cam0, cam1)
        np.testing.assert_array_equal(valid, valido)
        np.testing.assert_allclose(X, Xo, rtol=1e-12); np.testing.assert_allclose(rep, repo, rtol=1e-9, atol=1e-12)
    Xs, reps, vals = modules.triangulate(xy0[sel].astype(np.float32), xy1[sel].astype(np.float32), cam0, cam1)
    assert Xs.dtype == np.float32 and reps.dtype == np.float32 and vals.dtype == bool      # class follows the points
    e = ops.triangulate(np.zeros((0, 2)), np.zeros((0, 2)), cam0, cam1)
    assert e[0].shape == (0, 3) and e[1].shape == (0,) and e[2].shape == (0,)
    with pytest.raises(ValueError):
        ops.triangulate(xy0[:3], xy1[:2], cam0, cam1)


def test_angle_cheirality_reprojection(ctx):
    rng = np.random.default_rng(9)
    X = rng.normal(size=(1000, 3)) * 10
    p0 = synth.pose43(synth.rot_y(5.0), [0.3, 0.1, -1.0]); p1 = synth.pose43(synth.rot_y(-3.0), [1.0, 0.0, 0.5])
    ref = np.array([oracle.angle_deg(x, p0, p1) for x in X])
    for cm in (False, True):
        np.testing.assert_allclose(ops.angle_deg(X, p0, p1, col_major=cm), ref, rtol=1e-9)
    assert modules.calculateAngleDeg([0, 0, 0], synth.pose43(np.eye(3), [1, 0, 0]),
                                     synth.pose43(np.eye(3), [0, 1, 0])) == pytest.approx(90.0, abs=1e-9)
    cam = oracle.camera_matrix(p0, K)
    np.testing.assert_array_equal(modules.isInFrontOfCamera(cam, X), oracle.in_front(cam, X))
    xy = rng.uniform(0, 1000, (1000, 2))
    np.testing.assert_allclose(ops.reproj_error(xy, X, p0, K), oracle.reproj_error(xy, X, p0, K), rtol=1e-12)
    assert ops.angle_deg(np.zeros((0, 3)), p0, p1).shape == (0,)


# ---------------------------------------------------------------- P3P / RANSAC
def test_p3p_batch_matches_golden(ctx, golden):
    g = golden
    got = ops.p3p_batch(g["p3p_P"], g["p3p_f"])
    np.testing.assert_allclose(got, g["p3p_poses"], rtol=1e-8, atol=1e-8)
    P = np.array([[0, 0, 5.0], [1, 1, 6.0], [2, 2, 7.0]]); f = P / np.linalg.norm(P, axis=1, keepdims=True)
    assert not ops.p3p_batch(P[None], f[None]).any()              # collinear: zeros (old/p3p.m:56-58)
    m = modules.p3p(g["p3p_P"][0].T, g["p3p_f"][0].T)             # MATLAB 3 x 16 layout
    np.testing.assert_allclose(m[:, 0], g["p3p_poses"][0, 0, :3], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(m[:, 1:4], g["p3p_poses"][0, 0, 3:].reshape(3, 3), rtol=1e-8, atol=1e-8)


def test_p3p_batch_solves_the_geometry(ctx):
    """Evidence for the device P3P that shares no text with the oracle (the oracle's p3p.c and the kernel restate
    the same published algorithm line by line): Kneip's construction checked in numpy.  For random poses and point
    triples, (1) the true pose is among the four solutions, and (2) EVERY returned solution is a solution of the
    three-point problem -- a proper rotation whose camera sees each world point along its bearing -- unless it comes
    from a complex root (see below)."""
    rng = np.random.default_rng(77)
    n = 4000
    A = rng.normal(size=(n, 3, 3))
    Q = np.linalg.qr(A)[0]
    Q[:, :, 0] *= np.sign(np.linalg.det(Q))[:, None]          # proper rotations, camera -> world
    C = rng.normal(size=(n, 3)) * 3.0
    Pc = rng.uniform(-2, 2, (n, 3, 3)) + np.array([0, 0, 6.0])   # three points in front of the camera
    Pw = np.einsum("nij,nkj->nki", Q, Pc) + C[:, None, :]        # world = R_wc * cam + C
    f = Pc / np.linalg.norm(Pc, axis=2, keepdims=True)
    got = ops.p3p_batch(Pw, f)                                    # n x 4 x [C(3) R_wc(9)]
    Cs, Rs = got[:, :, :3], got[:, :, 3:].reshape(n, 4, 3, 3)
    err_true = np.maximum(np.abs(Cs - C[:, None]).max(2), np.abs(Rs - Q[:, None]).max((2, 3)))
    assert (err_true.min(1) < 1e-6).mean() > 0.999 and np.median(err_true.min(1)) < 1e-10
    # The quartic has two or four real roots; old/p3p.m:187-215 takes real() of a complex pair too, which gives the
    # same non-solution twice.  So: at least two of the four reproduce the three bearings (proper rotations), and what
    # does not is one of a pair of equal entries.  (Bars on the bulk: next to a double root both statements blur.)
    cam = np.einsum("nsji,nskj->nski", Rs, Pw[:, None] - Cs[:, :, None, :])      # R_wc^T (P - C)
    dirs = cam / np.linalg.norm(cam, axis=3, keepdims=True)
    bear = np.abs(dirs - f[:, None]).max((2, 3))
    solves = bear < 1e-5
    assert ((solves.sum(1) >= 2).mean() > 0.95) and np.median(bear[solves]) < 1e-10
    RtR = np.einsum("nsji,nsjk->nsik", Rs, Rs)
    assert np.abs(RtR - np.eye(3)).max((2, 3))[solves].max() < 1e-6
    assert np.abs(np.linalg.det(Rs)[solves] - 1.0).max() < 1e-6
    twin = np.zeros((n, 4), bool)
    for a_ in range(4):
        for b_ in range(4):
            if a_ != b_:
                twin[:, a_] |= np.abs(got[:, a_] - got[:, b_]).max(1) < 1e-9
    off = bear > 1e-3
    assert off.any() and twin[off].mean() > 0.95


@pytest.mark.parametrize("mode", [0, 1])
def test_ransac_identical_hypotheses_bit_exact(ctx, golden, mode):
    """Inlier sets, counts and MSAC costs under IDENTICAL hypotheses (the oracle's poses)."""
    g = golden
    r = ops.p3p_ransac(g["rs_xy"], g["rs_X"], K, num_hyp=64, max_reproj_err=2.0, seed=4095, mode=mode, min_inliers=4,
                       hyp_poses=g[f"rs{mode}_poses"], want_all=True)
    np.testing.assert_array_equal(r["counts"], g[f"rs{mode}_counts"])
    np.testing.assert_array_equal(r["costs"], g[f"rs{mode}_costs"])
    np.testing.assert_array_equal(r["inliers"], g[f"rs{mode}_inliers"])
    assert [r["best_hyp"], r["best_sol"], r["trials"], r["status"]] == list(g[f"rs{mode}_best"])
    assert r["num_inliers"] == int(g[f"rs{mode}_inliers"].sum())
    np.testing.assert_array_equal(r["R"], g[f"rs{mode}_R"])       # the winning pose is copied, not recomputed
    np.testing.assert_allclose(r["t"], -g[f"rs{mode}_R"].T @ g[f"rs{mode}_t"], rtol=1e-12)


@pytest.mark.parametrize("mode", [0, 1])
def test_ransac_seeded_stream_matches_oracle(ctx, golden, mode):
    """Same seed -> same Philox samples -> same hypotheses up to libm ulps -> same decisions."""
    g = golden
    r = ops.p3p_ransac(g["rs_xy"], g["rs_X"], K, num_hyp=64, max_reproj_err=2.0, seed=4095, mode=mode, min_inliers=4,
                       want_all=True)
    d = np.abs(r["poses"] - g[f"rs{mode}_poses"]).reshape(64 * 4, 12).max(1)
    assert np.median(d) < 1e-10 and (d < 1e-6).mean() > 0.95     # ill-conditioned triplets amplify libm ulps
    assert (r["counts"] == g[f"rs{mode}_counts"]).mean() > 0.98
    np.testing.assert_array_equal(r["inliers"], g[f"rs{mode}_inliers"])
    assert r["best_hyp"] == g[f"rs{mode}_best"][0]
    table = np.array([oracle.philox_sample3(4095, h, len(g["rs_xy"])) for h in range(64)])
    t = ops.p3p_ransac(g["rs_xy"], g["rs_X"], K, num_hyp=64, max_reproj_err=2.0, seed=1, mode=mode, min_inliers=4,
                       sample_idx=table, want_all=True)
    np.testing.assert_array_equal(t["poses"], r["poses"])         # explicit table == seeded stream
    for cm in (True,):
        c = ops.p3p_ransac(g["rs_xy"], g["rs_X"], K, num_hyp=64, max_reproj_err=2.0, seed=4095, mode=mode,
                           min_inliers=4, col_major=cm)
        np.testing.assert_array_equal(c["inliers"], r["inliers"]); np.testing.assert_array_equal(c["R"], r["R"])


def test_ransac_stress_shape_and_adaptive_stop(ctx):
    """BASELINE stress: 50k points x 2000 hypotheses; oracle scores the GPU's own hypotheses."""
    xy, X, R_cw, C, is_out = synth.pnp_scene(50000)
    r = ops.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj_err=2.0, seed=4095, mode=1, min_inliers=4, want_all=True)
    assert r["status"] == 0 and r["trials"] == 2000
    assert (r["inliers"] & ~is_out).sum() > 0.95 * (~is_out).sum() and (r["inliers"] & is_out).sum() < 50
    np.testing.assert_allclose(r["R"], R_cw, atol=2e-3); np.testing.assert_allclose(r["t"], C, atol=0.05)
    # bounded oracle work: 200 hypotheses; the GPU then scores the oracle's poses (identical hypotheses)
    o = oracle.p3p_ransac(xy, X, K, num_hyp=200, max_reproj=2.0, seed=4095, mode=1, min_inliers=4, want_all=True,
                          threads=8)
    s = ops.p3p_ransac(xy, X, K, num_hyp=200, max_reproj_err=2.0, seed=4095, mode=1, min_inliers=4,
                       hyp_poses=o["poses"], want_all=True)
    np.testing.assert_array_equal(s["counts"], o["counts"])
    np.testing.assert_array_equal(s["costs"], o["costs"])
    np.testing.assert_array_equal(s["inliers"], o["inliers"])
    assert (s["best_hyp"], s["best_sol"]) == (o["best_hyp"], o["best_sol"])
    assert (r["counts"][:200] == o["counts"]).mean() > 0.98        # own hypotheses: equal up to libm ulps
    for conf in (0.99, 0.5):
        a = ops.p3p_ransac(xy[:3000], X[:3000], K, num_hyp=500, max_reproj_err=2.0, confidence=conf, seed=7, mode=0)
        b = oracle.p3p_ransac(xy[:3000], X[:3000], K, num_hyp=500, max_reproj=2.0, confidence=conf, seed=7, mode=0)
        assert a["trials"] == b["trials"] < 500 and a["best_hyp"] == b["best_hyp"]
        np.testing.assert_array_equal(a["inliers"], b["inliers"])


def test_ransac_statuses(ctx):
    xy, X, *_ = synth.pnp_scene(100, noise=0.1, outlier_frac=0.0)
    few = ops.p3p_ransac(xy[:3], X[:3], K, num_hyp=10)
    assert few["status"] == 1 and not few["inliers"].any() and np.array_equal(few["R"], np.eye(3))
    junk_xy = np.random.default_rng(0).uniform(0, 300, (50, 2))
    junk = ops.p3p_ransac(junk_xy, X[:50], K, num_hyp=20, min_inliers=30)
    assert junk["status"] == 2 and not junk["inliers"].any()
    R, t, inl, status = modules.estimateWorldCameraPose(xy, X, modules.CameraParams(K), MaxNumTrials=300,
                                                        Confidence=99, MaxReprojectionError=2)
    assert status == 0 and inl.mean() > 0.9 and t.shape == (1, 3)   # a minimal-sample pose, no refit


# ---------------------------------------------------------------- the pipeline mirror
def test_process_frame_runs_on_a_synthetic_scene(ctx):
    """modules/VisualOdometry.m:264-346 end to end on a static scene: the camera does not move, so
    the pose step must return the identity pose and landmarks survive."""
    img, _, _ = synth.frame_pair(nv=240, nu=320, seed=51)
    Kc = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
    kp = modules.selectCandidateKeypoints(img, np.zeros((0, 2)), MinQuality=0.001, MinDistance=9, CandidatesToKeep=150)
    rng = np.random.default_rng(1)
    Z = rng.uniform(4, 30, len(kp))
    X = np.stack([(kp[:, 0] - Kc[0, 2]) / Kc[0, 0] * Z, (kp[:, 1] - Kc[1, 2]) / Kc[1, 1] * Z, Z], 1)
    st = modules._new_state()
    st.keypoints = kp.astype(np.float64); st.landmarks = X; st.reproError = np.zeros(len(kp))
    vo = modules.VisualOdometry(modules.CameraParams(Kc), img.shape, maxNumLandmarks=500, maxReprojectionError=2)
    st2, pose, status = vo.processFrame(img, img, st)
    assert status == 0
    np.testing.assert_allclose(pose[:3], np.eye(3), atol=1e-3); np.testing.assert_allclose(pose[3], 0, atol=2e-2)
    assert len(st2.landmarks) > 0.8 * len(kp) and len(st2.candidate_keypoints) > 0
    assert len(st2.candidate_first_poses) == len(st2.candidate_keypoints) == len(st2.candidate_time_indxs)
    st3, pose3, status3 = vo.processFrame(img, img, st2)          # candidates are now tracked and aged
    assert status3 == 0 and (st3.candidate_time_indxs <= -1).all()


@pytest.mark.parametrize("m", [1500, 300, 1024, 37])
@pytest.mark.parametrize("outlier_frac,mode", [(0.2, 1), (0.75, 0), (0.75, 1), (0.93, 1)])
def test_ransac_adaptive_run_in_two_stages(ctx, outlier_frac, mode, m):
    """With a confidence and more than 512 hypotheses the launcher scores the first 256 first and the rest only
    if the sequential stop was not reached among them: the stop inside the first stage (20 % outliers), beyond it
    (75 %: about 290 trials) and never (93 %) all give the oracle's trial count, best hypothesis and inlier set.
    Up to 1024 points a stage is ONE launch (k_pose_small: each block generates and scores its own hypotheses, the
    last one to finish scans); above, three."""
    xy, X, *_ = synth.pnp_scene(m, seed=77, outlier_frac=outlier_frac)
    a = ops.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj_err=2.0, confidence=0.99, seed=11, mode=mode, min_inliers=4)
    b = oracle.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj=2.0, confidence=0.99, seed=11, mode=mode, min_inliers=4, threads=8)
    assert a["trials"] == b["trials"] and a["status"] == b["status"]
    if m >= 300:
        assert a["status"] == 0
        assert (a["trials"] < 256) if outlier_frac == 0.2 else (a["trials"] > 256)
    else:                      # 37 points: with 93 % outliers there are fewer inliers than a model needs (status 2)
        assert a["status"] == (2 if outlier_frac > 0.9 else 0)
    assert (a["best_hyp"], a["best_sol"]) == (b["best_hyp"], b["best_sol"])
    np.testing.assert_array_equal(a["inliers"], b["inliers"])
    np.testing.assert_allclose(a["R"], b["R"], rtol=1e-9, atol=1e-12)


def test_short_pose_runs_between_other_users_of_the_work_buffer(ctx):
    """The pose step's work buffer is laid out per call (by the number of hypotheses) and shared with the essential-matrix
    run; the one-launch stage of short runs (k_pose_small) counts finished blocks in a word of its own, which must read
    zero whatever ran in between (a word inside the shared buffer did not: the last block never scanned and the run
    returned the record of an earlier one)."""
    rng = np.random.default_rng(5)
    xy, X, *_ = synth.pnp_scene(300, seed=91, outlier_frac=0.2)
    want = oracle.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj=2.0, confidence=0.99, seed=3, mode=1, min_inliers=4, threads=8)
    p0 = rng.uniform(50, 1100, (400, 2)); p1 = p0 + rng.normal(0, 2, p0.shape)
    for other in range(4):
        if other == 1:
            ops.essential_ransac(p0, p1, K, num_hyp=700, confidence=0.0, seed=1, ctx=ctx)
        elif other == 2:
            big, bigX, *_ = synth.pnp_scene(5000, seed=92, outlier_frac=0.3)
            ops.p3p_ransac(big, bigX, K, num_hyp=3000, max_reproj_err=2.0, confidence=0.0, seed=4, mode=1, ctx=ctx)
        elif other == 3:
            ops.p3p_ransac(xy, X, K, num_hyp=600, max_reproj_err=2.0, confidence=0.99, seed=9, mode=0, ctx=ctx)
        got = ops.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj_err=2.0, confidence=0.99, seed=3, mode=1, min_inliers=4, ctx=ctx)
        assert (got["status"], got["trials"], got["best_hyp"], got["best_sol"]) == (want["status"], want["trials"], want["best_hyp"], want["best_sol"]), other
        np.testing.assert_array_equal(got["inliers"], want["inliers"])


def test_ransac_adaptive_run_still_scores_everything_when_asked(ctx):
    """Per-hypothesis outputs (counts, costs, poses) of an adaptive run cover all hypotheses, not just the first stage."""
    xy, X, *_ = synth.pnp_scene(1200, seed=78, outlier_frac=0.2)
    a = ops.p3p_ransac(xy, X, K, num_hyp=1500, max_reproj_err=2.0, confidence=0.99, seed=12, mode=1, min_inliers=4, want_all=True)
    b = oracle.p3p_ransac(xy, X, K, num_hyp=1500, max_reproj=2.0, confidence=0.99, seed=12, mode=1, min_inliers=4, want_all=True,
                          threads=8)
    assert a["trials"] == b["trials"] < 256 and a["counts"].shape == b["counts"].shape == (1500, 4)
    assert (a["counts"] == b["counts"]).mean() > 0.98 and (a["counts"][300:] > 0).any()   # own hypotheses: equal up to libm ulps
    s = ops.p3p_ransac(xy, X, K, num_hyp=1500, max_reproj_err=2.0, confidence=0.99, seed=12, mode=1, min_inliers=4,
                       hyp_poses=b["poses"], want_all=True)
    np.testing.assert_array_equal(s["counts"], b["counts"])
    np.testing.assert_array_equal(s["costs"], b["costs"])
    assert s["trials"] == b["trials"] and s["best_hyp"] == b["best_hyp"]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("confidence", [0.0, 0.99])
def test_ransac_early_out_changes_no_result(ctx, mode, confidence):
    """Runs of more than 512 hypotheses score the hypotheses behind the first 256 with an early out: a pose whose
    cost (count) can no longer strictly beat the first stage's record is dropped from the remaining points (k_score,
    ScoreArgs::bound).  Record, trial count, pose and inlier set must be those of the full scoring, which a caller
    gets by asking for the per-hypothesis outputs (want_all)."""
    for m, frac, hyp, seed, min_inl in ((25000, 0.2, 2000, 3, 4), (6000, 0.5, 1500, 4, 30), (9000, 0.05, 700, 5, 4),
                                        (5000, 0.93, 900, 6, 4), (1500, 0.3, 3000, 7, 4)):
        xy, X, *_ = synth.pnp_scene(m, seed=80 + seed, outlier_frac=frac)
        kw = dict(num_hyp=hyp, max_reproj_err=2.0, confidence=confidence, seed=seed, mode=mode, min_inliers=min_inl)
        full = ops.p3p_ransac(xy, X, K, want_all=True, **kw)
        fast = ops.p3p_ransac(xy, X, K, **kw)
        for key in ("status", "best_hyp", "best_sol", "trials", "num_inliers"):
            assert fast[key] == full[key], (key, m, frac, hyp)
        np.testing.assert_array_equal(fast["R"], full["R"]); np.testing.assert_array_equal(fast["t"], full["t"])
        np.testing.assert_array_equal(fast["inliers"], full["inliers"])
        # the hypotheses of the full run, scored by the fast path: still the same record
        again = ops.p3p_ransac(xy, X, K, hyp_poses=full["poses"], **kw)
        assert (again["best_hyp"], again["best_sol"], again["trials"]) == (full["best_hyp"], full["best_sol"], full["trials"])
        np.testing.assert_array_equal(again["inliers"], full["inliers"])
