"""Do two builds of the library give the same bits?  python scripts/ab_bits.py libA.so libB.so
Each library tracks the bench's 50k keypoints (and 3k keypoints with a 21x21 / 31x25 window) in a process of its own."""
import os, subprocess, sys
import numpy as np
if len(sys.argv) == 3 and sys.argv[1] != "--child":
    outs = []
    for lib in sys.argv[1:3]:
        out = "gpurun_out/bits_%s.npz" % os.path.basename(lib)
        subprocess.run([sys.executable, __file__, "--child", out], env=dict(os.environ, VO_HIP_LIB=os.path.abspath(lib)), check=True)
        outs.append(np.load(out))
    same = all(np.array_equal(outs[0][k], outs[1][k], equal_nan=True) for k in outs[0].files)
    print("identical bits:", same, {k: int((outs[0][k] != outs[1][k]).sum()) for k in outs[0].files})
    if not same:
        print("largest difference:", {k: float(np.nanmax(np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)))) for k in outs[0].files if k[0] != "v"})
    sys.exit(0 if same else 1)
sys.path.insert(0, ".")
import ctypes as C
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload, _p
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, 25000, 25000, n_frames=2, num_hyp=8, ahead=False)
wl.build(1, 1)
res = {}
def track(n, wu, wv):
    ctx.call("vo_dev_klt_track", 0, 1, _p(wl.pa[0]), _p(wl.pa[1]), n, wu, wv, 50, C.c_double(0.01), C.c_double(2.0),
             _p(wl.qa[0]), _p(wl.qa[1]), _p(wl.va), _p(wl.sa), None)
    wl.sync()
for tag, (wu, wv) in (("41", (41, 41)), ("21", (21, 21)), ("31", (31, 31)), ("51", (51, 51)), ("35x31", (35, 31)), ("33x29", (33, 29)),
                      ("45x41", (45, 41)), ("43x37", (43, 37)), ("7", (7, 7)), ("19x21", (19, 21))):
    n = 50000 if tag == "41" else 3000
    track(n, wu, wv)
    res["q" + tag] = wl.qa[:, :n].cpu().numpy(); res["v" + tag] = wl.va[:n].cpu().numpy(); res["s" + tag] = wl.sa[:n].cpu().numpy()
np.savez(sys.argv[2], **res)
