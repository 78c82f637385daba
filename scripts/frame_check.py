"""Device-resident frame loop vs the oracle's processFrame restatement, frame by frame."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle, vo_frame
from visual_odometry_amd import synth
from visual_odometry_amd.frame import FrameLoop

K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames, poses, lift = synth.plane_sequence(F, K)
u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=150)
st = vo_frame.new_state()
st.keypoints = np.stack([u, v], 1).astype(np.float64); st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1])
st.reproError = np.zeros(len(u))
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=cap)
chain = FrameLoop(K, frames[0].shape, maxNumLandmarks=cap)
chain.set_state(st, frames[0])
for k in range(1, F):
    loop.set_state(st, frames[k - 1], frame_no=k - 1)          # same input state on both sides
    t0 = time.perf_counter()
    gpose, gstatus, ginfo = loop.processFrame(frames[k])
    t1 = time.perf_counter()
    g = loop.get_state()
    st2, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, K, k, maxNumLandmarks=cap)
    same_n = (len(g.landmarks) == len(st2.landmarks), len(g.candidate_keypoints) == len(st2.candidate_keypoints))
    msg = f"frame {k}: status {gstatus}/{status} n {len(g.landmarks)}/{len(st2.landmarks)} c {len(g.candidate_keypoints)}/{len(st2.candidate_keypoints)}"
    msg += f" pose diff {np.abs(gpose - pose).max():.2e} gpu {1e3 * (t1 - t0):.2f} ms info {ginfo['n_tracked']}/{info['tracked']} prom {ginfo['n_promoted']}/{info['promoted']} new {ginfo['n_new']}/{info['new']}"
    if all(same_n):
        msg += f" | kp {np.abs(g.keypoints - st2.keypoints).max():.2e} L {np.abs(g.landmarks - st2.landmarks).max():.2e} rep {np.abs(g.reproError - st2.reproError).max():.2e}"
        if len(g.candidate_keypoints):
            msg += f" ck {np.abs(g.candidate_keypoints - st2.candidate_keypoints).max():.2e} cf {np.abs(g.candidate_first_keypoints - st2.candidate_first_keypoints).max():.2e} cp {np.abs(g.candidate_first_poses - st2.candidate_first_poses).max():.2e} ct {np.abs(g.candidate_time_indxs - st2.candidate_time_indxs).max()}"
    print(msg)
    cpose, cstatus, cinfo = chain.processFrame(frames[k])
    print(f"   chain: n {cinfo['n_landmarks']} c {cinfo['n_candidates']} pose diff vs oracle chain {np.abs(cpose - pose).max():.2e}")
    st = st2
