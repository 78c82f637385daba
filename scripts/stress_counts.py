"""Counts of the device-resident frame loop at the stress size, frame by frame (landmarks, candidates, promoted, new,
overflow), with the bench leg's candidate capacity (25 000: full at the start) and with room (40 000)."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from visual_odometry_amd import _lib, modules, synth
from visual_odometry_amd.frame import FrameLoop
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(25, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
st = synth.stress_state(K, lift, modules._new_state, 25000, 25000, max_age=20)
for cap in (25000, 40000):
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=25000, maxCandidates=cap, CandidatesToKeep=5000, MinDistance=1.5, angularThreshold=12.0, maxTemporalRecall=24)
    loop.set_state(st, dev[0])
    rows = []
    for k in range(1, 25):
        _, _, inf = loop.processFrame(dev[k])
        rows.append((inf["n_landmarks"], inf["n_candidates"], inf["n_promoted"], inf["n_new"], inf["overflow"]))
    print("maxCandidates", cap, "(landmarks, candidates, promoted, new, overflow) per frame:")
    print(" ".join(str(r) for r in rows))
    loop.close()
