import ctypes as C, sys, torch
sys.path.insert(0, '.')
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload, _p
ctx = _lib.Context(0)
for n in (25000, 1000):
    wl = FrameWorkload(ctx, n, n)
    om = torch.zeros(400, dtype=torch.float64, device=wl.dev)
    for it in range(6):
        wl.ctx.call("vo_dev_select_finish", 0, _p(wl.ex[0]), _p(wl.ex[1]), wl.n_exist, _p(wl.sel_u), _p(wl.sel_v), _p(wl.sel_lin), _p(om), _p(wl.sel_cnt))
        wl.ctx.sync()
        o = om.cpu().numpy()
        print(n, "ticks(10ns):", o[200:208], "rounds,nl,m:", o[210:213], "count", int(wl.sel_cnt.item()))
        k1 = it + 1
        wl.prefetch(k1 % 3, (k1 % wl.n_frames) + 1); wl.ctx.sync()
