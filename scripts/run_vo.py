"""MATLAB-free end-to-end run: bootstrap (modules/bootstrap.m) on two frames, then the device-resident
processFrame loop (modules/VisualOdometry.m:264-346) over the rest of a synthetic non-planar sequence,
like module_tests/VisualOdometry_test.m:30-103 does on a dataset.  Prints the trajectory against the
truth (monocular: compared after fixing the scale on the bootstrap baseline).

    python scripts/run_vo.py [n_frames] [nv nu]
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from visual_odometry_amd import modules, synth
from visual_odometry_amd.frame import FrameLoop

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nv, nu = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (376, 1241)
K = synth.KITTI_K if (nv, nu) == (376, 1241) else np.array([[300.0, 0, nu / 2], [0, 300.0, nv / 2], [0, 0, 1.0]])
frames, poses, lift = synth.relief_sequence(n_frames, K, nv=nv, nu=nu, depth=20.0, relief=4.0, step=(0.3, 0.02, 0.1))
K1 = K.copy(); K1[0, 2] += 1; K1[1, 2] += 1                       # the mirror speaks MATLAB's 1-based pixels
t0 = time.perf_counter()
kp, X, rep, pose = modules.bootstrap(frames[0], frames[2], modules.CameraParams(K1))
print(f"bootstrap: {len(kp)} landmarks, {1e3 * (time.perf_counter() - t0):.1f} ms (first call: includes warm-up)")
scale = np.linalg.norm(poses[2][3])                                # fix the monocular scale on the bootstrap baseline
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=300)
best = np.argsort(rep)[:300]
st = modules._new_state()
st.keypoints, st.landmarks, st.reproError = kp[best] - 1, X[best], rep[best]
st.candidate_first_poses = np.zeros((0, 4, 3))
loop.set_state(st, frames[2])
t0 = time.perf_counter()
for k in range(3, n_frames):
    p, status, info = loop.processFrame(frames[k])
    err = np.linalg.norm(p[3] * scale - poses[k][3])
    print(f"frame {k:3d}: status {status} landmarks {info['n_landmarks']:4d} candidates {info['n_candidates']:4d} "
          f"promoted {info['n_promoted']:3d} |C - truth| = {err:.3f} m of {np.linalg.norm(poses[k][3]):.2f} m travelled")
dt = time.perf_counter() - t0
print(f"{n_frames - 3} frames in {1e3 * dt:.1f} ms = {(n_frames - 3) / dt:.0f} frames/s (host images uploaded per frame)")
