"""The frame loop from a selected start state (bench.py's frame_loop legs), for A/B switches read at create:
    VO_FRAME_WIDE_MIN=1024 python scripts/frame_sizes.py --landmarks 2000 --keep 300"""
import argparse, statistics, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from visual_odometry_amd import _lib, modules, ops, synth
from visual_odometry_amd.frame import FrameLoop
ap = argparse.ArgumentParser()
ap.add_argument("--landmarks", type=int, default=2000)
ap.add_argument("--keep", type=int, default=300)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--passes", type=int, default=10)
a = ap.parse_args()
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(a.frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
ctx = _lib.Context(0)
kp = ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), min_distance=7, min_quality=0.001, keep=a.landmarks,
                           index_base=0, ctx=ctx).astype(np.float64)
st = modules._new_state()
st.keypoints = kp; st.landmarks = lift(kp[:, 0], kp[:, 1]); st.reproError = np.zeros(len(kp))
st.candidate_first_poses = np.zeros((0, 4, 3))
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=a.landmarks, CandidatesToKeep=a.keep, ctx=ctx)
times = []
for _ in range(a.passes):
    loop.set_state(st, dev[0]); ctx.sync()
    t0 = time.perf_counter()
    pending = 0
    for k in range(1, a.frames):
        loop.submit(dev[k], dev[k + 1] if k + 1 < a.frames else None); pending += 1
        if pending == 2:
            res = loop.collect(); pending -= 1
    while pending:
        res = loop.collect(); pending -= 1
    times.append(time.perf_counter() - t0)
n = a.frames - 1
print("ms/frame best %.4f median %.4f" % (1e3 * min(times) / n, 1e3 * statistics.median(times[1:]) / n), res[2])
