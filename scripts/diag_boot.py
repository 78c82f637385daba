import numpy as np, sys
sys.path.insert(0,'.')
from visual_odometry_amd import synth, modules, _lib
from visual_odometry_amd.frame import FrameLoop
K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
frames, poses, lift = synth.relief_sequence(7, K)
K1 = K.copy(); K1[0, 2] += 1; K1[1, 2] += 1
kp, X, rep, pose, info = modules.bootstrap(frames[0], frames[2], modules.CameraParams(K1), return_info=True)
st = modules._new_state()
st.keypoints = kp - 1; st.landmarks = X; st.reproError = rep
st.candidate_first_poses = np.zeros((0, 4, 3))
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=400)
sel = np.argsort(rep)[:400]
st.keypoints, st.landmarks, st.reproError = st.keypoints[sel], st.landmarks[sel], st.reproError[sel]
loop.set_state(st, frames[2])
Ct = poses[2][3]; scale = 1.0/np.linalg.norm(Ct)
for k in range(3,7):
    p, status, fi = loop.processFrame(frames[k])
    c_est, c_true = p[3], poses[k][3]*scale
    s = loop.get_state()
    print(k, status, fi["n_tracked"], fi["n_inliers"], fi["trials_used"], fi["n_landmarks"], fi["n_candidates"], "cos %.4f"%(c_est@c_true/(np.linalg.norm(c_est)*np.linalg.norm(c_true))), "kp sum %.6f"%s.keypoints.sum())
