import sys
sys.path.insert(0, ".")
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, 25000, 25000, n_frames=2, num_hyp=8, ahead=False)
wl.select(1); wl.sync()
print("selected", int(wl.sel_cnt.item()), "existing", wl.n_exist, "ex range", float(wl.ex[0].min()), float(wl.ex[0].max()), float(wl.ex[1].min()), float(wl.ex[1].max()))
ctx.timer_start()
for _ in range(20): wl.select(1)
print("select ms", ctx.timer_stop_ms() / 20)
