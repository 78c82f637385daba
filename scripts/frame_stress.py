"""GPU-only driver for the device-resident frame loop at the stress size (bench.py `frame_loop.at_50k_keypoints`):
25 000 landmarks at the cap + 25 000 seeded candidates, up to 5000 new candidates a frame.
    python scripts/frame_stress.py [--frames 24] [--passes 5] [--depth 2] [--sigma 2.0] [--angle 12] [--recall 24] [--max-age 20] [--landmarks 25000] [--candidates 25000] [--keep 5000] [--min-distance 1.5] [--counts]"""
import argparse, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from visual_odometry_amd import _lib, modules, synth
from visual_odometry_amd.frame import FrameLoop
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--depth", type=int, default=2, help="frames in flight")
ap.add_argument("--sigma", type=float, default=2.0, help="blur of the plane's texture")
ap.add_argument("--angle", type=float, default=12.0, help="angularThreshold, degrees")
ap.add_argument("--recall", type=int, default=24, help="maxTemporalRecall")
ap.add_argument("--max-age", type=int, default=20, help="the seeded candidates were first seen 1 .. max-age frames ago")
ap.add_argument("--landmarks", type=int, default=25000, help="maxNumLandmarks = landmarks the state starts with")
ap.add_argument("--candidates", type=int, default=25000, help="maxCandidates = seeded candidates")
ap.add_argument("--keep", type=int, default=5000, help="CandidatesToKeep")
ap.add_argument("--min-distance", type=float, default=1.5)
ap.add_argument("--counts", action="store_true", help="one more pass that prints the state's counts frame by frame")
a = ap.parse_args()
n_frames, depth = a.frames, a.depth
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15), sigma=a.sigma)
ctx = _lib.Context(0)
st = synth.stress_state(K, lift, modules._new_state, a.landmarks, a.candidates, max_age=a.max_age)
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=a.landmarks, maxCandidates=a.candidates, CandidatesToKeep=a.keep, MinDistance=a.min_distance,
                 angularThreshold=a.angle, maxTemporalRecall=a.recall, ctx=ctx)
for rep in range(a.passes):
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
if a.counts:
    loop.set_state(st, dev[0])
    tot = 0
    for k in range(1, n_frames):
        pose, status, info = loop.processFrame(dev[k], dev[k + 1] if k + 1 < n_frames else None)
        tot += info["n_tracked"] + info["n_candidates"]
        print(k, {x: info[x] for x in ("n_landmarks", "n_candidates", "n_tracked", "n_promoted", "n_new")})
    print("landmarks tracked + candidates held, mean per frame:", tot / (n_frames - 1))
