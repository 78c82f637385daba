"""GPU-only driver for the device-resident frame loop at the stress size (bench.py `frame_loop.at_50k_keypoints`):
25 000 landmarks at the cap + 25 000 seeded candidates, 5000 new candidates a frame.
    python scripts/frame_stress.py [n_frames] [passes] [frames in flight]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from visual_odometry_amd import _lib, modules, synth
from visual_odometry_amd.frame import FrameLoop
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
ctx = _lib.Context(0)
st = synth.stress_state(K, lift, modules._new_state)
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=25000, maxCandidates=25000, CandidatesToKeep=5000, MinDistance=1.5,
                 angularThreshold=6.0, maxTemporalRecall=16, ctx=ctx)
for rep in range(reps):
    loop.set_state(st, dev[0]); ctx.sync()
    t0 = time.perf_counter()
    pending = 0
    for k in range(1, n_frames):
        nxt = dev[k + 1] if k + 1 < n_frames else None
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
