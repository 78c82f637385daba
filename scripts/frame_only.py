"""GPU-only driver for profiling the device-resident frame loop (vo_frame_process)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from oracle import oracle, vo_frame
from visual_odometry_amd import _lib, synth
from visual_odometry_amd.frame import FrameLoop
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=300)
st = vo_frame.new_state()
st.keypoints = np.stack([u, v], 1).astype(np.float64); st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1])
st.reproError = np.zeros(len(u))
ctx = _lib.Context(0)
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, ctx=ctx)
for rep in range(3):
    loop.set_state(st, dev[0]); ctx.sync()
    t0 = time.perf_counter()
    for k in range(1, n_frames):
        pose, status, info = loop.processFrame(dev[k])
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {1e3 * dt / (n_frames - 1):.3f} ms/frame  {info}")
