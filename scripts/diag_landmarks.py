"""Diagnostic: depth consistency of the frame loop's landmarks on the rendered relief scene (truth known)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from visual_odometry_amd import modules, synth
from visual_odometry_amd.frame import FrameLoop

import os
nv, nu, n = 240, 320, 27
K = np.array([[300.0, 0, nu / 2], [0, 300.0, nv / 2], [0, 0, 1.0]])
scene = [float(v) for v in os.environ.get("VO_SCENE", "20,4,0.3,0.02,0.1").split(",")]
frames, poses, lift = synth.relief_sequence(n, K, nv=nv, nu=nu, depth=scene[0], relief=scene[1], step=tuple(scene[2:5]))
quiet = bool(os.environ.get("VO_QUIET"))
K1 = K.copy(); K1[0, 2] += 1; K1[1, 2] += 1
kp, X, rep, pose = modules.bootstrap(frames[0], frames[2], modules.CameraParams(K1), MinNumLandmarks=50, MaxDepth=200.0,
                                     FilterSize=3, MinQuality=0.005)
scale = np.linalg.norm(poses[2][3]) / np.linalg.norm(pose[3])
print("bootstrap", len(kp), "scale", scale, "pose C", pose[3] * scale, "truth", poses[2][3])


def ratios(L):
    uv = (K @ L.T).T
    u0, v0 = uv[:, 0] / uv[:, 2], uv[:, 1] / uv[:, 2]
    ok = (u0 > 0) & (u0 < nu - 1) & (v0 > 0) & (v0 < nv - 1) & (L[:, 2] > 0)
    T = lift(u0[ok], v0[ok])
    return T[:, 2] / (L[ok, 2] * scale), ok


r, ok = ratios(X)
print("bootstrap landmarks: Z_true / Z_est  median %.4f  p10 %.4f p90 %.4f" % (np.median(r), np.percentile(r, 10), np.percentile(r, 90)))
best = np.argsort(rep, kind="stable")[:500]
st = modules._new_state()
st.keypoints, st.landmarks, st.reproError = kp[best] - 1.0, X[best], rep[best]
st.candidate_first_poses = np.zeros((0, 4, 3))
opts = dict(maxNumLandmarks=500, maxReprojectionError=2.0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    opts[k] = float(v)
loop = FrameLoop(K, frames[0].shape, **opts)
loop.set_state(st, frames[2])
for k in range(3, n):
    p, status, info = loop.processFrame(frames[k])
    s = loop.get_state()
    r, ok = ratios(s.landmarks)
    err = np.linalg.norm(p[3] * scale - poses[k][3])
    if quiet and k not in (5, 10, 15, 20, 26):
        continue
    print("frame %2d status %d N %3d promoted %3d inl %3d  C err %.3f m | Z_true/Z_est median %.3f p10 %.3f p90 %.3f (%d in view 0)" % (
        k, status, info["n_landmarks"], info["n_promoted"], info["n_inliers"], err, np.median(r), np.percentile(r, 10), np.percentile(r, 90), ok.sum()))
