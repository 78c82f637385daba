"""Experiment: does the order of the keypoints in a k_lk launch matter (L2 locality per XCD)?
Same keypoints, same launch, three input orders: as drawn (random), sorted by Morton key of the position,
and Morton-sorted then dealt so that blocks b, b+8, b+16 ... (one XCD under round-robin placement) walk one
contiguous eighth of the sorted list.   usage: python scripts/klt_order.py [n] [reps]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, n, 16, n_frames=2, num_hyp=8, ahead=False)
wl.build(1, 1)
base = wl.pl.clone()


def morton(u, v, shift=4):
    x = (u.astype(np.int64) >> shift); y = (v.astype(np.int64) >> shift)
    key = np.zeros_like(x)
    for b in range(12):
        key |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return key


p = base.cpu().numpy()
order_m = np.argsort(morton(p[0], p[1]), kind="stable")
b = np.arange(n)
per = (n + 7) // 8
deal = np.minimum((b % 8) * per + b // 8, n - 1)          # block b -> rank in the sorted list (n divisible by 8 here)
order_m6 = np.argsort(morton(p[0], p[1], 6), kind="stable")
rowmajor = np.lexsort((p[0], (p[1].astype(np.int64) >> 5)))     # 32-row bands, left to right inside a band
orders = {"as drawn": np.arange(n), "morton 16px": order_m, "morton 16px, dealt per XCD": order_m[deal],
          "morton 64px": order_m6, "32-row bands": rowmajor, "32-row bands, dealt per XCD": rowmajor[deal]}
for _ in range(200):          # clocks up before the first order is measured
    wl.track_landmarks(0, 1)
wl.sync()
ref = None
for name, o in list(orders.items()) + list(orders.items()):
    wl.pl.copy_(torch.from_numpy(np.ascontiguousarray(p[:, o])).to(wl.pl.device))
    for _ in range(5):
        wl.track_landmarks(0, 1)
    wl.sync()
    ctx.timer_start()
    for _ in range(reps):
        wl.track_landmarks(0, 1)
    ms = ctx.timer_stop_ms() / reps
    q = wl.ql.cpu().numpy()
    back = np.empty_like(q); back[:, o] = q
    if ref is None:
        ref = back
    same = np.array_equal(ref, back, equal_nan=True)
    print(f"{name:24s} {ms:.4f} ms/launch   results identical to the first order: {same}")
