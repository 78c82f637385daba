"""GPU-only driver for profiling the device-resident frame loop (vo_frame_process).
    python scripts/frame_only.py [n_frames] [max_landmarks] [new_candidates_per_frame] [look_ahead 0|1] [frames in flight]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from visual_odometry_amd import _lib, modules, ops, synth
from visual_odometry_amd.frame import FrameLoop
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_land = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_new = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ahead = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 2
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
ctx = _lib.Context(0)
kp = ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), min_distance=7, min_quality=0.001, keep=n_land, index_base=0,
                           ctx=ctx).astype(np.float64)
st = modules._new_state()
st.keypoints = kp; st.landmarks = lift(kp[:, 0], kp[:, 1]); st.reproError = np.zeros(len(kp))
st.candidate_first_poses = np.zeros((0, 4, 3))
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=n_land, CandidatesToKeep=n_new, ctx=ctx)
for rep in range(3):
    loop.set_state(st, dev[0]); ctx.sync()
    t0 = time.perf_counter()
    pending = 0
    for k in range(1, n_frames):
        nxt = dev[k + 1] if ahead and k + 1 < n_frames else None
        if depth <= 1:
            pose, status, info = loop.processFrame(dev[k], nxt)
        else:
            loop.submit(dev[k], nxt); pending += 1
            if pending == depth:
                pose, status, info = loop.collect(); pending -= 1
    while pending:
        pose, status, info = loop.collect(); pending -= 1
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {1e3 * dt / (n_frames - 1):.3f} ms/frame  {info}")
