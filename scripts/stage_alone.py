"""Each stage of a frame step on its own (a sync after every call), for `rocprofv3 --kernel-trace --stats`:
per-kernel durations with nothing else on the chip.  usage: python3 scripts/stage_alone.py [n_landmarks] [reps]"""
import sys
sys.path.insert(0, '.')
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, n, n)
ctx.sync()
for it in range(reps):
    wl.select_finish(); ctx.sync()
    k1 = it + 1
    wl.prefetch(k1 % 3, (k1 % wl.n_frames) + 1); ctx.sync()
    wl.pose(); ctx.sync()
    wl.triangulate(); ctx.sync()
print("done")
