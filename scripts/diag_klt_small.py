"""KLT on a small image (320 x 240, 4 levels: the top level is smaller than the 41 x 41 window) against the oracle."""
import numpy as np, sys
sys.path.insert(0, '.')
from visual_odometry_amd import synth, ops
from oracle import oracle
K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
frames, poses, lift = synth.relief_sequence(7, K)
u, v, _, _ = oracle.select_candidates(frames[2], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=400)
pts = np.stack([u, v], 1).astype(np.float64)
out, valid, score = ops.klt_track(frames[2], frames[3], pts, levels=4, block=(41, 41), index_base=0)
qu, qv, ov, *_ = oracle.klt_track(frames[2], frames[3], pts[:, 0], pts[:, 1], levels=4, wu=41, wv=41)
both = valid & ov
d = np.hypot(out[:, 0] - qu, out[:, 1] - qv)
print("n", len(pts), "valid gpu", valid.sum(), "oracle", ov.sum(), "mismatch", (valid != ov).sum(), "max diff", d[both].max(), "n>1e-3:", (d[both] > 1e-3).sum())
bad = np.flatnonzero((d > 1e-3) & both)
for i in bad[:10]:
    print(i, pts[i], out[i], (qu[i], qv[i]))
