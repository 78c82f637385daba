"""CPU restatement of modules/bootstrap.m:24-135 on top of the C oracle.  TEST INFRASTRUCTURE ONLY
(oracle/vo_oracle.h); PARITY UNPINNED: estimateEssentialMatrix / relativeCameraPose /
triangulateMultiview are closed toolbox functions, defined here as MSAC around Nister's five-point solver (N10b,
five_point.c; Solver='eight-point': the normalised eight-point algorithm, N10, essential.c), the textbook four-fold
decomposition with a cheirality vote, and the
two-view DLT of the frame step.  Zero-based (u, v) coordinates, K in the same origin."""
import numpy as np

from . import oracle

DEFAULTS = dict(MinDepth=0.0, MaxDepth=5000.0, MinNumLandmarks=50, MinQuality=0.01, FilterSize=5,   # :13-21
                uniformFeaturesPercentage=100.0, MaxNumTrials=2000, Confidence=99.5, MaxDistance=0.09,  # :76-79
                levels=4, max_bidir=2.0, block=41, max_iters=50,                                     # KLTTracker() defaults
                seed=1023, max_attempts=20, threads=4, Solver="five-point")


def decompose_essential(E):
    """The four (R, t) with E = [t]x R (x1 = R x0 + t), in a fixed order."""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    Ra, Rb, t = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    return [(Ra, t), (Ra, -t), (Rb, t), (Rb, -t)]


def relative_camera_pose(E, K, p0, p1):
    """relativeCameraPose (:93-94): -> (orientation R_cw, location C (unit), validPointsFraction)."""
    n = len(p0)
    pose0 = np.vstack([np.eye(3), np.zeros((1, 3))])
    best = None
    for R, t in decompose_essential(E):
        pose1 = np.vstack([R, (-(R.T @ t)).reshape(1, 3)])
        _, _, _, ok = oracle.triangulate(p0, p1, np.repeat(pose0[None], n, 0), pose1, K)
        c = int(ok.sum())
        if best is None or c > best[0]:
            best = (c, pose1)
    return best[1][:3], best[1][3], best[0] / max(n, 1)


def bootstrap(img0, img1, K, **kw):
    """-> (keypoints N x 2 in img1, landmarks N x 3, reproError N, pose 4 x 3, info)."""
    o = dict(DEFAULTS); o.update(kw)
    none = np.zeros(0, np.float32)
    nv, nu = img0.shape
    cap = nu * nv // 4 + 16
    u, v, _, _ = oracle.select_candidates(img0, none, none, min_distance=0.0, min_quality=o["MinQuality"],
                                          filter_size=o["FilterSize"], keep=cap)                     # :30-32
    n_keep = int(round(len(u) * o["uniformFeaturesPercentage"] / 100.0))                              # :36-38
    if n_keep < len(u):
        u, v, _, _ = oracle.select_candidates(img0, none, none, min_distance=0.0, min_quality=o["MinQuality"],
                                              filter_size=o["FilterSize"], keep=n_keep)
    qu, qv, valid, *_ = oracle.klt_track(img0, img1, u.astype(np.float64), v.astype(np.float64), levels=o["levels"],
                                         max_bidir=o["max_bidir"], wu=o["block"], wv=o["block"], max_iters=o["max_iters"],
                                         threads=o["threads"])                                       # :57-60
    m0 = np.stack([u, v], 1).astype(np.float64)[valid]
    m1 = np.stack([qu, qv], 1)[valid]
    pose0 = np.vstack([np.eye(3), np.zeros((1, 3))])
    for attempt in range(o["max_attempts"]):                                                          # :74
        solve = oracle.essential_ransac5 if o["Solver"] == "five-point" else oracle.essential_ransac
        r = solve(m0, m1, K, num_hyp=o["MaxNumTrials"], max_distance=o["MaxDistance"], confidence=o["Confidence"] / 100.0,
                  seed=o["seed"] + attempt, min_inliers=5 if o["Solver"] == "five-point" else 8, threads=o["threads"])  # :76-79
        inl = r["inliers"]
        if r["status"] != 0 or inl.sum() < o["MinNumLandmarks"]:                                      # :83-86
            continue
        a, b = m0[inl], m1[inl]
        R1, t1, frac = relative_camera_pose(r["E"], K, a, b)                                          # :93-94
        if inl.sum() * frac < o["MinNumLandmarks"] or not np.isfinite(R1).all() or not np.isfinite(t1).all():
            continue                                                                                  # :97-100
        pose = np.vstack([R1, t1.reshape(1, 3)])
        X, rep, _, ok = oracle.triangulate(a, b, np.repeat(pose0[None], len(a), 0), pose, K)          # :120-121
        ok = ok & (X[:, 2] > o["MinDepth"]) & (X[:, 2] <= o["MaxDepth"])                              # :124-125
        if ok.sum() >= o["MinNumLandmarks"]:                                                          # :128
            info = dict(corners=len(u), tracked=int(valid.sum()), inliers=int(inl.sum()), attempts=attempt + 1,
                        valid_fraction=frac, trials=r["trials"])
            return b[ok], X[ok], rep[ok], pose, info                                                  # :132-135
    raise RuntimeError("bootstrap: no attempt produced enough landmarks")
