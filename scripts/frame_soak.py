"""Soak of the frame loop at the stress size: the same sequence many times, every pass must end in the same state
(counts, poses and a checksum of the state arrays) -- the one-wave compactions publish counts between workgroups and
the merge sorts on the device; a race would show as a pass that differs.
    python scripts/frame_soak.py [passes] [frames]"""
import hashlib, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from visual_odometry_amd import _lib, modules, synth
from visual_odometry_amd.frame import FrameLoop
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
ctx = _lib.Context(0)
st = synth.stress_state(K, lift, modules._new_state, max_age=20)
dev = [torch.from_numpy(f).to("cuda:0") for f in frames]
loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=25000, maxCandidates=25000, CandidatesToKeep=5000, MinDistance=1.5,
                 angularThreshold=12.0, maxTemporalRecall=24, ctx=ctx)
seen = {}
for rep in range(passes):
    loop.set_state(st, dev[0])
    h = hashlib.sha256()
    pending = 0
    for k in range(1, n_frames):
        loop.submit(dev[k], dev[k + 1] if k + 1 < n_frames else None); pending += 1
        if pending == 2:
            pose, status, info = loop.collect(); pending -= 1
            h.update(np.ascontiguousarray(pose).tobytes()); h.update(repr(sorted(info.items())).encode())
    while pending:
        pose, status, info = loop.collect(); pending -= 1
        h.update(np.ascontiguousarray(pose).tobytes()); h.update(repr(sorted(info.items())).encode())
    g = loop.get_state()
    for f in ("landmarks", "keypoints", "reproError", "candidate_keypoints", "candidate_first_keypoints", "candidate_time_indxs"):
        h.update(np.ascontiguousarray(getattr(g, f)).tobytes())
    seen.setdefault(h.hexdigest(), []).append(rep)
print("passes", passes, "distinct outcomes", len(seen), {k[:12]: len(v) for k, v in seen.items()})
sys.exit(0 if len(seen) == 1 else 1)
