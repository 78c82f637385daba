"""GPU-only driver for profiling k_lk: device API, N keypoints, R launches."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, n, 16, n_frames=2, num_hyp=8, max_iters=mi, ahead=False)
wl.build(1, 1)
wl.track_landmarks(0, 1); wl.sync()
ctx.counters(reset=True)
ctx.timer_start()
for _ in range(reps):
    wl.track_landmarks(0, 1)
ms = ctx.timer_stop_ms() / reps
it, kp, su, ce = ctx.counters()
fl = 1681 * (11 * it + 24 * su) / reps
print(f"n={n} {ms:.4f} ms/launch  {n/ms/1e3:.2f} Mkp/s  iters/kp={it/kp:.2f} cells/kp={ce/kp:.2f}  {fl/ms/1e9:.2f} TFLOP/s  valid={wl.vl.float().mean().item():.4f}")
