"""numpy-facing wrappers over the host-pointer entry points of libvo_hip.so (include/vo_hip.h).

Every function takes numpy arrays, calls ONE C-ABI entry point and returns numpy arrays.  The
memory layout is taken from the image / first array: Fortran-ordered input means MATLAB layout
(`col_major=1`), anything else is made C-contiguous (`col_major=0`).  `index_base` says whether
pixel coordinates are MATLAB's (1, the default of `modules.py`) or zero-based.

No CPU fallback: `_lib.Context` raises when the shared library or a HIP device is missing.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Context, KltOpts, Layout, RansacInfo, RansacOpts, SelOpts, ptr

SCORERS = {"mineigen": _lib.SCORER_MINEIGEN, "harris": _lib.SCORER_HARRIS}


def _is_f(a):
    return a.ndim == 2 and a.flags["F_CONTIGUOUS"] and not a.flags["C_CONTIGUOUS"]


def _img(img):
    img = np.asarray(img)
    if img.ndim != 2 or img.dtype != np.uint8:
        raise TypeError("image must be a 2-D uint8 array (grayscale)")
    if _is_f(img):
        return img, 1
    return np.ascontiguousarray(img), 0


def _mat(a, col_major, dtype, cols):
    """n x cols array in the requested memory order."""
    a = np.asarray(a, dtype=dtype)
    if a.ndim == 1 and a.size == cols:
        a = a.reshape(1, cols)
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError(f"expected an N x {cols} array, got shape {a.shape}")
    return np.asfortranarray(a) if col_major else np.ascontiguousarray(a)


def _empty(n, cols, col_major, dtype):
    return np.empty((n, cols), dtype=dtype, order="F" if col_major else "C")


def klt_track(img1, img2, pts, levels=4, max_bidir_err=2.0, block=(41, 41), max_iters=50, eps=0.01,
              index_base=1, ctx=None):
    """KLTTracker.track (modules/KLTTracker.m:22-41) -> (points, validity, scores)."""
    ctx = ctx or Context.default()
    img1, cm = _img(img1)
    img2, cm2 = _img(img2)
    if img1.shape != img2.shape:
        raise ValueError("images differ in size")
    if cm2 != cm:
        img2 = np.asfortranarray(img2) if cm else np.ascontiguousarray(img2)
    pts = np.asarray(pts)
    dt = np.float32 if pts.dtype == np.float32 else np.float64
    if pts.size == 0:  # modules/KLTTracker.m:33-36
        return np.empty((0, 2), np.float64), np.empty((0,), bool), np.empty((0,), np.float64)
    p = _mat(pts, cm, dt, 2)
    n = p.shape[0]
    out = _empty(n, 2, cm, dt)
    valid = np.zeros(n, np.uint8)
    scores = np.zeros(n, dt)
    o = KltOpts(int(levels), float(max_bidir_err), int(block[0]), int(block[1]), int(max_iters), float(eps))
    H, W = img1.shape
    ctx.call("vo_klt_track", ptr(img1), ptr(img2), H, W, Layout(cm, index_base), ptr(p), n, int(dt == np.float64),
             C.byref(o), ptr(out), ptr(valid), ptr(scores))
    return out, valid.astype(bool), scores


def select_candidates(img, existing, min_distance=7.0, min_quality=0.001, filter_size=5, keep=100,
                      scorer="mineigen", kappa=0.08, index_base=1, ctx=None, want_all=False):
    """selectCandidateKeypoints (modules/selectCandidateKeypoints.m:1-36) -> M x 2 float32 [x y]."""
    ctx = ctx or Context.default()
    img, cm = _img(img)
    existing = np.asarray(existing)
    dt = np.float64 if existing.dtype == np.float64 else np.float32
    ne = 0 if existing.size == 0 else existing.shape[0]
    e = _mat(existing, cm, dt, 2) if ne else np.zeros((1, 2), dt)
    keep = int(keep)
    out = np.zeros(max(keep, 1) * 2, np.float32)
    lin = np.zeros(max(keep, 1), np.int32)
    met = np.zeros(max(keep, 1), np.float64)
    m = C.c_int(0)
    o = SelOpts(float(min_distance), float(min_quality), int(filter_size), keep, SCORERS[scorer], float(kappa))
    H, W = img.shape
    ctx.call("vo_select_candidates", ptr(img), H, W, Layout(cm, index_base), ptr(e), ne, int(dt == np.float64),
             C.byref(o), ptr(out), ptr(lin), ptr(met), C.byref(m))
    M = m.value
    xy = out[:2 * M].reshape((M, 2), order="F" if cm else "C").copy(order="F" if cm else "C")
    if want_all:
        return xy, lin[:M].copy(), met[:M].copy()
    return xy


def corner_response(img, scorer="harris", filter_size=9, kappa=0.08, ctx=None):
    """old/harris.m:1-25 (scorer='harris') or the Shi-Tomasi metric of detectMinEigenFeatures."""
    ctx = ctx or Context.default()
    img, cm = _img(img)
    H, W = img.shape
    out = np.empty((H, W), np.float64, order="F" if cm else "C")
    ctx.call("vo_corner_response", ptr(img), H, W, Layout(cm, 0), SCORERS[scorer], int(filter_size),
             C.c_double(kappa), ptr(out))
    return out


def select_keypoints_greedy(scores, num, r, index_base=1, ctx=None):
    """old/selectKeypoints.m:1-16 -> 2 x num int32 [row; col]."""
    ctx = ctx or Context.default()
    scores = np.asarray(scores, np.float64)
    cm = 1 if _is_f(scores) else 0
    if not cm:
        scores = np.ascontiguousarray(scores)
    H, W = scores.shape
    out = np.zeros((2, int(num)), np.int32, order="F")
    ctx.call("vo_select_keypoints_greedy", ptr(scores), H, W, Layout(cm, index_base), int(num), int(r), ptr(out))
    return out


def _pose(p, cm):
    p = np.asarray(p, np.float64)
    if p.shape != (4, 3):
        raise ValueError("pose must be 4 x 3 [R; t]")
    return np.asfortranarray(p) if cm else np.ascontiguousarray(p)


def _K(K):
    K = np.ascontiguousarray(K, np.float64)
    if K.shape != (3, 3):
        raise ValueError("K must be 3 x 3")
    return K


def triangulate_candidates(first_xy, curr_xy, first_poses, curr_pose, K, first_pose_idx=None, col_major=False,
                           index_base=1, ctx=None):
    """modules/VisualOdometry.m:87-113 batched -> (X C x 3, reproj C, angle_deg C, valid C)."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    a = _mat(first_xy, cm, np.float64, 2)
    b = _mat(curr_xy, cm, np.float64, 2)
    Cn = a.shape[0]
    if b.shape[0] != Cn:
        raise ValueError("first_xy and curr_xy differ in length")
    fp = np.asarray(first_poses, np.float64).reshape(-1, 4, 3)
    P = fp.shape[0]
    # P blocks of 12 doubles, each block in the call's layout
    fpm = np.ascontiguousarray(np.transpose(fp, (0, 2, 1))) if cm else np.ascontiguousarray(fp)
    idx = None if first_pose_idx is None else np.ascontiguousarray(first_pose_idx, np.int32)
    X = _empty(Cn, 3, cm, np.float64)
    rep = np.zeros(Cn); ang = np.zeros(Cn); val = np.zeros(Cn, np.uint8)
    # the pixel coordinates enter the DLT as they are (K uses the same pixel convention)
    p1, Kc = _pose(curr_pose, cm), _K(K)   # keep the converted arrays alive across the call
    ctx.call("vo_triangulate_candidates", Layout(cm, index_base), ptr(a), ptr(b), ptr(fpm), P, ptr(idx), Cn,
             ptr(p1), ptr(Kc), ptr(X), ptr(rep), ptr(ang), ptr(val))
    return X, rep, ang, val.astype(bool)


def triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2, col_major=False, ctx=None):
    """The toolbox call as the reference makes it (modules/VisualOdometry.m:97-101, modules/candidateTriangulation.m:47-51,
    processFrame.m:55-59): N x 2 point pairs and two 4 x 3 camera matrices -> (worldPoints N x 3, reprojectionErrors N,
    validIndex N).  One C-ABI call (vo_triangulate_points); the batched form over per-candidate first poses is
    triangulate_candidates."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    a = _mat(matchedPoints1, cm, np.float64, 2)
    b = _mat(matchedPoints2, cm, np.float64, 2)
    n = a.shape[0]
    if b.shape[0] != n:
        raise ValueError("matchedPoints1 and matchedPoints2 differ in length")
    c0, c1 = _pose(camMatrix1, cm), _pose(camMatrix2, cm)      # 4 x 3, the call's layout
    X = _empty(n, 3, cm, np.float64)
    rep = np.zeros(n); val = np.zeros(n, np.uint8)
    ctx.call("vo_triangulate_points", Layout(cm, 1), ptr(a), ptr(b), n, ptr(c0), ptr(c1), ptr(X), ptr(rep), ptr(val))
    return X, rep, val.astype(bool)


def angle_deg(landmarks, pose0, pose1, col_major=False, ctx=None):
    """utils/calculateAngleDeg.m:1-11 batched over landmarks."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    X = _mat(landmarks, cm, np.float64, 3)
    out = np.zeros(X.shape[0])
    p0, p1 = _pose(pose0, cm), _pose(pose1, cm)
    ctx.call("vo_angle_deg", Layout(cm, 1), ptr(X), X.shape[0], ptr(p0), ptr(p1), ptr(out))
    return out


def in_front(cam_mat, points, col_major=False, ctx=None):
    """modules/isInFrontOfCamera.m:1-12."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    X = _mat(points, cm, np.float64, 3)
    out = np.zeros(X.shape[0], np.uint8)
    cam = _pose(cam_mat, cm)
    ctx.call("vo_in_front", Layout(cm, 1), ptr(cam), ptr(X), X.shape[0], ptr(out))
    return out.astype(bool)


def reproj_error(keypoints, landmarks, pose, K, col_major=False, ctx=None):
    """modules/VisualOdometry.m:48-61."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    xy = _mat(keypoints, cm, np.float64, 2)
    X = _mat(landmarks, cm, np.float64, 3)
    if xy.shape[0] != X.shape[0]:
        raise ValueError("keypoints and landmarks differ in length")
    out = np.zeros(X.shape[0])
    p, Kc = _pose(pose, cm), _K(K)
    ctx.call("vo_reproj_error", Layout(cm, 1), ptr(xy), ptr(X), X.shape[0], ptr(p), ptr(Kc), ptr(out))
    return out


def p3p_ransac(img_pts, world_pts, K, num_hyp=2000, max_reproj_err=2.0, confidence=0.0, seed=4095, mode=0,
               min_inliers=0, sample_idx=None, hyp_poses=None, col_major=False, want_all=False, ctx=None):
    """Pose step (modules/VisualOdometry.m:285-289; old/ransacLocalization.m) ->
    dict(R, t, inliers, status, best_hyp, best_sol, trials, num_inliers[, counts, costs, poses])."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    xy = _mat(img_pts, cm, np.float64, 2) if np.size(img_pts) else np.zeros((0, 2))
    X = _mat(world_pts, cm, np.float64, 3) if np.size(world_pts) else np.zeros((0, 3))
    M = xy.shape[0]
    if X.shape[0] != M:
        raise ValueError("image and world points differ in length")
    H = int(num_hyp)
    o = RansacOpts(H, float(max_reproj_err), float(confidence), int(seed), int(mode), int(min_inliers))
    si = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32).reshape(H, 3)
    hp = None if hyp_poses is None else np.ascontiguousarray(hyp_poses, np.float64).reshape(H, 4, 12)
    R = np.zeros((3, 3), order="F" if cm else "C"); t = np.zeros(3); inl = np.zeros(max(M, 1), np.uint8)
    info = RansacInfo()
    cnt = np.zeros((H, 4), np.int32) if want_all else None
    cost = np.zeros((H, 4), np.uint64) if want_all else None
    poses = np.zeros((H, 4, 12)) if want_all else None
    Kc = _K(K)
    ctx.call("vo_p3p_ransac", Layout(cm, 1), ptr(xy), ptr(X), M, ptr(Kc), C.byref(o), ptr(si), ptr(hp), ptr(R),
             ptr(t), ptr(inl), C.byref(info), ptr(cnt), ptr(cost), ptr(poses))
    out = dict(R=np.ascontiguousarray(R), t=t, inliers=inl[:M].astype(bool), status=info.status,
               best_hyp=info.best_hyp, best_sol=info.best_sol, trials=info.trials_used, num_inliers=info.num_inliers)
    if want_all:
        out.update(counts=cnt, costs=cost, poses=poses)
    return out


def essential_ransac(pts0, pts1, K, num_hyp=2000, max_distance=0.09, confidence=0.995, seed=1023, min_inliers=8,
                     sample_idx=None, hyp_E=None, col_major=False, want_all=False, ctx=None):
    """Relative-pose step of the bootstrap (modules/bootstrap.m:76-79): MSAC around the normalised
    eight-point algorithm -> dict(E, inliers, status, best_hyp, trials, num_inliers[, counts, costs, Es])."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    p0 = _mat(pts0, cm, np.float64, 2) if np.size(pts0) else np.zeros((0, 2))
    p1 = _mat(pts1, cm, np.float64, 2) if np.size(pts1) else np.zeros((0, 2))
    M = p0.shape[0]
    if p1.shape[0] != M:
        raise ValueError("point sets differ in length")
    H = int(num_hyp)
    o = RansacOpts(H, float(max_distance), float(confidence), int(seed), 1, int(min_inliers))
    si = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32).reshape(H, 8)
    he = None if hyp_E is None else np.ascontiguousarray(hyp_E, np.float64).reshape(H, 9)
    E = np.zeros((3, 3), order="F" if cm else "C"); inl = np.zeros(max(M, 1), np.uint8)
    info = RansacInfo()
    cnt = np.zeros(H, np.int32) if want_all else None
    cost = np.zeros(H, np.uint64) if want_all else None
    Es = np.zeros((H, 3, 3)) if want_all else None
    Kc = _K(K)
    ctx.call("vo_essential_ransac", Layout(cm, 1), ptr(p0), ptr(p1), M, ptr(Kc), C.byref(o), ptr(si), ptr(he), ptr(E),
             ptr(inl), C.byref(info), ptr(cnt), ptr(cost), ptr(Es))
    out = dict(E=np.ascontiguousarray(E), inliers=inl[:M].astype(bool), status=info.status, best_hyp=info.best_hyp,
               trials=info.trials_used, num_inliers=info.num_inliers)
    if want_all:
        out.update(counts=cnt, costs=cost, Es=Es)
    return out


def essential_ransac_five(pts0, pts1, K, num_hyp=2000, max_distance=0.09, confidence=0.995, seed=1023, min_inliers=5,
                          sample_idx=None, col_major=False, want_all=False, ctx=None):
    """The relative-pose step of the bootstrap (modules/bootstrap.m:76-79) around Nister's five-point solver -- works on
    planar scenes, which the eight-point algorithm does not -> dict(E, inliers, status, best_hyp, trials,
    num_inliers[, counts, costs, Es]); the per-hypothesis outputs have ten slots per sample, best_hyp is a slot index."""
    ctx = ctx or Context.default()
    cm = 1 if col_major else 0
    p0 = _mat(pts0, cm, np.float64, 2) if np.size(pts0) else np.zeros((0, 2))
    p1 = _mat(pts1, cm, np.float64, 2) if np.size(pts1) else np.zeros((0, 2))
    M = p0.shape[0]
    if p1.shape[0] != M:
        raise ValueError("point sets differ in length")
    H = int(num_hyp)
    o = RansacOpts(H, float(max_distance), float(confidence), int(seed), 1, int(min_inliers))
    si = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32).reshape(H, 5)
    E = np.zeros((3, 3), order="F" if cm else "C"); inl = np.zeros(max(M, 1), np.uint8)
    info = RansacInfo()
    cnt = np.zeros(10 * H, np.int32) if want_all else None
    cost = np.zeros(10 * H, np.uint64) if want_all else None
    Es = np.zeros((10 * H, 3, 3)) if want_all else None
    Kc = _K(K)
    ctx.call("vo_essential_ransac_five", Layout(cm, 1), ptr(p0), ptr(p1), M, ptr(Kc), C.byref(o), ptr(si), ptr(E),
             ptr(inl), C.byref(info), ptr(cnt), ptr(cost), ptr(Es))
    out = dict(E=np.ascontiguousarray(E), inliers=inl[:M].astype(bool), status=info.status, best_hyp=info.best_hyp,
               trials=info.trials_used, num_inliers=info.num_inliers)
    if want_all:
        out.update(counts=cnt, costs=cost, Es=Es)
    return out


def p3p_batch(world_pts, bearings, ctx=None):
    """old/p3p.m:41-217 batched: n x 3 x 3 points / unit bearings -> n x 4 x 12 [C(3) R(9)]."""
    ctx = ctx or Context.default()
    P = np.ascontiguousarray(world_pts, np.float64).reshape(-1, 3, 3)
    f = np.ascontiguousarray(bearings, np.float64).reshape(-1, 3, 3)
    n = P.shape[0]
    out = np.zeros((n, 4, 12))
    ctx.call("vo_p3p_batch", ptr(P), ptr(f), n, ptr(out))
    return out
