"""Scratch GPU check: vo_klt_track (HIP) against the oracle on one synthetic pair; prints timing."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from visual_odometry_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
img1, img2, truth = synth.frame_pair(shift=(3.3, -2.6), zoom=0.004, noise=1.0)
pu, pv = synth.keypoints(n)
t0 = time.time()
qu, qv, valid, score, fb, iters = oracle.klt_track(img1, img2, pu, pv, threads=8)
t_cpu = time.time() - t0
tu, tv = truth(pu, pv)
err = np.hypot(qu - tu, qv - tv)
print(f"oracle: {t_cpu:.2f}s valid={valid.mean():.3f} err(valid) median={np.median(err[valid]):.4f} "
      f"p95={np.percentile(err[valid], 95):.4f} iters/pt={iters.mean():.1f}")
if "--cpu" in sys.argv:
    sys.exit(0)

ctx = _lib.Context(0)
pts = np.stack([pu, pv], 1).astype(np.float64).copy()
out = np.zeros_like(pts); ov = np.zeros(n, np.uint8); osc = np.zeros(n)
o = _lib.KltOpts(4, 2.0, 41, 41, 50, 0.01)
lay = _lib.Layout(0, 0)
H, W = img1.shape
for rep in range(3):
    t0 = time.time()
    ctx.call("vo_klt_track", _lib.ptr(img1), _lib.ptr(img2), H, W, lay, _lib.ptr(pts), n, 1, C.byref(o),
             _lib.ptr(out), _lib.ptr(ov), _lib.ptr(osc))
    print(f"gpu host-API call {rep}: {(time.time() - t0) * 1e3:.2f} ms")
d = np.hypot(out[:, 0] - qu, out[:, 1] - qv)
both = valid & ov.astype(bool)
print(f"gpu: valid={ov.mean():.3f} valid-mismatch={(valid != ov.astype(bool)).sum()} "
      f"max|gpu-oracle|(both valid)={d[both].max():.5f} px  p99={np.percentile(d[both], 99):.5f} "
      f"score maxdiff={np.abs(osc - score)[both].max():.5f}")
print("counters", ctx.counters())
