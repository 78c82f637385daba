"""What would an ordering of the keypoints buy k_lk?  50k keypoints: as generated (random), sorted by image tile,
sorted and dealt out so that the workgroups one XCD receives (blockIdx % 8) cover one eighth of the image."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, n, 16, n_frames=2, num_hyp=8, ahead=False)
wl.build(1, 1)
p0 = wl.pl.clone()
u, v = p0[0].cpu().numpy(), p0[1].cpu().numpy()
key = (v // tile).astype(np.int64) * 4096 + (u // tile).astype(np.int64)
srt = np.argsort(key, kind="stable")
chunk = (n + 7) // 8
b = np.arange(n)
deal = srt[np.minimum((b % 8) * chunk + b // 8, n - 1)] if n % 8 == 0 else srt
for name, order in (("random", np.arange(n)), ("tile-sorted", srt), ("sorted+xcd", deal), ("random", np.arange(n))):
    wl.pl.copy_(p0[:, torch.from_numpy(order).to(p0.device)])
    for _ in range(60):
        wl.track_landmarks(0, 1)
    wl.sync()
    ctx.timer_start()
    for _ in range(reps):
        wl.track_landmarks(0, 1)
    ms = ctx.timer_stop_ms() / reps
    print(f"{name:12s} n={n} {ms:.4f} ms/launch valid={wl.vl.float().mean().item():.4f}")
