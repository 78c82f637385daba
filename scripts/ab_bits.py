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
for tag, blk in (("41", 41), ("21", 21), ("31", 31), ("51", 51)):
    wl.block = blk
    n = 50000 if blk == 41 else 3000
    wl.track(0, 1, wl.pa[:, :n], wl.qa[:, :n], wl.va[:n], wl.sa[:n]); wl.sync()
    res["q" + tag] = wl.qa[:, :n].cpu().numpy(); res["v" + tag] = wl.va[:n].cpu().numpy(); res["s" + tag] = wl.sa[:n].cpu().numpy()
np.savez(sys.argv[2], **res)
