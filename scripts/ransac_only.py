import sys
sys.path.insert(0, ".")
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, n, 16, n_frames=2, num_hyp=2000, ahead=False)
wl.pose(); wl.sync()
ctx.counters(reset=True)
ctx.timer_start()
for _ in range(10):
    wl.pose()
ms = ctx.timer_stop_ms() / 10
c = ctx.counters()
print(f"M={n} ransac {ms:.4f} ms  counter3/launch={c[3]/10:.0f}  frac of pairs={(c[3]/10)/(2000*4*n):.4f}  inliers={int(wl.rs_inl.sum())}")
