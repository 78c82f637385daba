"""Host time of vo_frame_submit and the loop's rate with four frames in flight.
    python scripts/submit_cost.py [host]      host: hand over host images (numpy) instead of device images"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from visual_odometry_amd import _lib, modules, ops, synth
from visual_odometry_amd.frame import FrameLoop
K = synth.KITTI_K
n_frames = 24
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
ctx = _lib.Context(0)
kp = ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), min_distance=7, min_quality=0.001, keep=300, index_base=0, ctx=ctx).astype(np.float64)
st = modules._new_state(); st.keypoints = kp; st.landmarks = lift(kp[:, 0], kp[:, 1]); st.reproError = np.zeros(len(kp)); st.candidate_first_poses = np.zeros((0, 4, 3))
host = len(sys.argv) > 1 and sys.argv[1] == "host"
dev = [np.ascontiguousarray(f) for f in frames] if host else [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, ctx=ctx)
for rep in range(4):
    loop.set_state(st, dev[0]); ctx.sync()
    ts = []
    k = 1
    while k + 4 <= n_frames:
        t0 = time.perf_counter()
        for j in range(4):
            loop.submit(dev[k + j], dev[k + j + 1] if k + j + 1 < n_frames else None)
        t1 = time.perf_counter()
        for j in range(4): loop.collect()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) / 4, (t2 - t0) / 4)); k += 4
    print("host time per vo_frame_submit %.3f ms; 4 frames submitted then collected: %.3f ms per frame" % (np.median([a for a, b in ts]) * 1e3, np.median([b for a, b in ts]) * 1e3))
