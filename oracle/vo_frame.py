"""CPU restatement of the per-frame step, modules/VisualOdometry.m:264-346 (processFrame) with the
method candidateTriangulation (:63-151), on top of the C oracle (oracle.py).

TEST INFRASTRUCTURE ONLY (see oracle/vo_oracle.h): used by tests/ to check the device-resident frame
loop (vo_frame_* in include/vo_hip.h).  PARITY UNPINNED: the reference holds no recorded states.

Coordinates are the oracle's internal zero-based (u = column, v = row); K must use the same origin.
State arrays are float64; the reference's single/double drift ("dtype drift") only enters
through the single cast of the existing points in the selection step (:330-331), which is kept.
"""
from types import SimpleNamespace

import numpy as np

from . import oracle

DEFAULTS = dict(angularThreshold=1.0, maxTemporalRecall=10, maxNumLandmarks=300, maxReprojectionError=3.0,  # :22-25
                levels=4, max_bidir=2.0, block=41, max_iters=50,                                          # :33-37
                trials=5000, confidence=0.99, pose_tol=2.0,                                               # :285-289
                min_quality=0.001, filter_size=5, min_distance=7.0, keep=100,                             # :332-335
                seed=1023, threads=4)


def new_state():
    return SimpleNamespace(landmarks=np.zeros((0, 3)), keypoints=np.zeros((0, 2)), reproError=np.zeros(0),
                           candidate_keypoints=np.zeros((0, 2)), candidate_first_keypoints=np.zeros((0, 2)),
                           candidate_first_poses=np.zeros((0, 4, 3)), candidate_time_indxs=np.zeros(0, np.int64))


def _track(prev_img, curr_img, pts, o):
    if len(pts) == 0:
        return np.zeros((0, 2)), np.zeros(0, bool)
    qu, qv, valid, *_ = oracle.klt_track(prev_img, curr_img, pts[:, 0], pts[:, 1], levels=o["levels"],
                                         max_bidir=o["max_bidir"], wu=o["block"], wv=o["block"],
                                         max_iters=o["max_iters"], threads=o["threads"])
    return np.stack([qu, qv], 1), valid


def pose_from_rt(R_cw, t_cw):
    """[R_WC; t_WC] of :318: orientation = R_cw, location C = -R_cw^T t_cw (summed in index order)."""
    C = np.array([-((R_cw[0, c] * t_cw[0] + R_cw[1, c] * t_cw[1]) + R_cw[2, c] * t_cw[2]) for c in range(3)])
    return np.vstack([R_cw, C.reshape(1, 3)])


def process_frame(prev_img, curr_img, prev, K, frame_no, **kw):
    """-> (curr_state, curr_pose 4x3, pose_status, info).  `frame_no` (1 for the first processed
    frame) offsets the RANSAC seed: seed + frame_no."""
    o = dict(DEFAULTS); o.update(kw)
    # ---- :278-281 track the landmarks' keypoints
    tk, val = _track(prev_img, curr_img, prev.keypoints, o)
    vk, vl = tk[val], prev.landmarks[val]
    # ---- :285-289 pose
    r = oracle.p3p_ransac(vk, vl, K, num_hyp=o["trials"], max_reproj=o["pose_tol"], confidence=o["confidence"],
                          seed=o["seed"] + frame_no, mode=1, min_inliers=4, threads=o["threads"])
    pose = pose_from_rt(r["R"], r["t"])
    # ---- :298-315 reprojection filter, strict <
    err = oracle.reproj_error(vk, vl, pose, K) if len(vk) else np.zeros(0)
    keep = err < o["maxReprojectionError"]
    landmarks, keypoints, repro = list(vl[keep]), list(vk[keep]), list(err[keep])
    # ---- :322 -> :63-151 candidate triangulation and merge
    ct, cval = _track(prev_img, curr_img, prev.candidate_keypoints, o)
    idx = np.flatnonzero(cval)
    if idx.size:
        X, rep, ang, ok = oracle.triangulate(prev.candidate_first_keypoints[idx], ct[idx],
                                             prev.candidate_first_poses[idx], pose, K)
    ck, cfk, cfp, cti = [], [], [], []
    n_promoted = n_replaced = 0
    for j, i in enumerate(idx):
        if repro:                                                  # :89-90, before this iteration's edits
            mi = int(np.argmax(repro)); mx = repro[mi]
        else:
            mi, mx = -1, -np.inf
        if not (ok[j] and rep[j] <= o["maxReprojectionError"]):     # :106-111
            continue
        if ang[j] > o["angularThreshold"]:                          # :113
            if len(landmarks) >= o["maxNumLandmarks"] and rep[j] <= mx:   # :116-122
                del landmarks[mi]; del keypoints[mi]; del repro[mi]
                n_replaced += 1
            if len(landmarks) < o["maxNumLandmarks"]:               # :123-127
                landmarks.append(X[j]); keypoints.append(ct[i]); repro.append(rep[j])
                n_promoted += 1
        elif prev.candidate_time_indxs[i] > -o["maxTemporalRecall"]:  # :130-139
            ck.append(ct[i]); cfk.append(prev.candidate_first_keypoints[i])
            cfp.append(prev.candidate_first_poses[i]); cti.append(prev.candidate_time_indxs[i] - 1)
    cs = new_state()
    cs.landmarks = np.array(landmarks).reshape(-1, 3); cs.keypoints = np.array(keypoints).reshape(-1, 2)
    cs.reproError = np.array(repro).reshape(-1)
    ck = np.array(ck).reshape(-1, 2)
    # ---- :330-335 new candidates away from every tracked point (single cast, :330-331)
    both = np.vstack([cs.keypoints, ck]).astype(np.float32)
    nu_, nv_, _, _ = oracle.select_candidates(curr_img, both[:, 0], both[:, 1], min_distance=o["min_distance"],
                                              min_quality=o["min_quality"], filter_size=o["filter_size"], keep=o["keep"])
    new = np.stack([nu_, nv_], 1).astype(np.float64)
    # ---- :338-345 append
    cs.candidate_keypoints = np.vstack([ck, new])
    cs.candidate_first_keypoints = np.vstack([np.array(cfk).reshape(-1, 2), new])
    cs.candidate_first_poses = np.concatenate([np.array(cfp).reshape(-1, 4, 3), np.repeat(pose[None], len(new), 0)])
    cs.candidate_time_indxs = np.concatenate([np.array(cti, np.int64), -np.ones(len(new), np.int64)])
    info = dict(tracked=int(val.sum()), inliers=int(r["inliers"].sum()), trials=r["trials"], promoted=n_promoted,
                replaced=n_replaced, new=len(new))
    return cs, pose, r["status"], info
