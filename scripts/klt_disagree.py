"""Diagnostic: where do the HIP tracker and the oracle disagree on validity, and why?  For each parity scenario of
tests/test_gpu_parity.py: the points whose validity differs, with both sides' forward-backward errors and positions."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from oracle import oracle
from visual_odometry_amd import _lib, synth
from visual_odometry_amd._lib import ptr

ctx = _lib.Context(0)


def klt_dev(img1, img2, pu, pv, levels, wu, wv, max_iters, max_bidir):
    nv, nu = img1.shape
    a = torch.from_numpy(np.ascontiguousarray(img1)).cuda(); b = torch.from_numpy(np.ascontiguousarray(img2)).cuda()
    ctx.call("vo_dev_frame_build", 0, ptr(a), nu, nv, levels)
    ctx.call("vo_dev_frame_build", 1, ptr(b), nu, nv, levels)
    p = torch.from_numpy(np.stack([pu, pv]).astype(np.float64)).cuda()
    q = torch.zeros_like(p); n = p.shape[1]
    valid = torch.zeros(n, dtype=torch.uint8, device="cuda"); score = torch.zeros(n, dtype=torch.float32, device="cuda")
    fb = torch.zeros(n, dtype=torch.float32, device="cuda")
    ctx.call("vo_dev_klt_track", 0, 1, ptr(p[0]), ptr(p[1]), n, wu, wv, max_iters, C.c_double(0.01), C.c_double(max_bidir),
             ptr(q[0]), ptr(q[1]), ptr(valid), ptr(score), ptr(fb))
    ctx.sync()
    q = q.cpu().numpy()
    return q[0], q[1], valid.cpu().numpy().astype(bool), score.cpu().numpy(), fb.cpu().numpy()


def report(name, img1, img2, pu, pv, levels, wu, wv, max_iters, max_bidir, threads=8):
    g = klt_dev(img1, img2, pu, pv, levels, wu, wv, max_iters, max_bidir)
    o = oracle.klt_track(img1, img2, pu, pv, levels=levels, max_bidir=max_bidir, wu=wu, wv=wv, max_iters=max_iters, threads=threads)
    both = g[2] & o[2]
    d = np.hypot(g[0] - o[0], g[1] - o[1])
    dis = np.flatnonzero(g[2] != o[2])
    nv, nu = img1.shape
    print(f"== {name}: n={len(pu)} both={both.sum()} max|dq|(both)={d[both].max() if both.any() else -1:.5f} disagree={len(dis)} "
          f"score diff={np.abs(g[3]-o[3])[both].max() if both.any() else -1:.2e}")
    for i in dis[:12]:
        print(f"   p=({pu[i]:.2f},{pv[i]:.2f}) gvalid={g[2][i]} ovalid={o[2][i]} gfb={g[4][i]:.4f} ofb={o[4][i]:.4f} "
              f"|dq|={d[i]:.4f} q_o=({o[0][i]:.2f},{o[1][i]:.2f}) border_dist={min(pu[i], pv[i], nu-1-pu[i], nv-1-pv[i]):.1f}")


img1, img2, _ = synth.frame_pair(shift=(3.3, -2.6), zoom=0.004, noise=1.0)
pu, pv = synth.keypoints(50000)
report("kitti 1500", img1, img2, pu[:1500], pv[:1500], 4, 41, 41, 50, 2.0)
for wu, wv, levels in [(7, 7, 3), (21, 21, 3), (23, 19, 2), (35, 31, 4), (41, 41, 4), (45, 41, 3), (41, 45, 3), (51, 51, 4), (55, 55, 2), (3, 55, 2)]:
    a, b, _ = synth.frame_pair(nv=150, nu=210, seed=11, shift=(1.7, -1.2), zoom=0.003)
    u, v = synth.keypoints(400, nv=150, nu=210, seed=12, border=0.0)
    report(f"window {wu}x{wv} L{levels}", a, b, u, v, levels, wu, wv, 30, 1.5)
for nv, nu in [(40, 48), (33, 200), (200, 37), (24, 24)]:
    a, b, _ = synth.frame_pair(nv=nv, nu=nu, seed=13, shift=(0.8, 0.6), margin=16)
    u, v = synth.keypoints(64, nv=nv, nu=nu, seed=14, border=2.0)
    report(f"small {nv}x{nu}", a, b, u, v, 3, 31, 31, 30, 2.0)
for nv, nu in [(8, 8), (8, 60), (90, 8), (9, 11)]:
    a, b, _ = synth.frame_pair(nv=nv, nu=nu, seed=19, shift=(0.4, 0.3), margin=8)
    u, v = synth.keypoints(40, nv=nv, nu=nu, seed=20, border=0.0)
    report(f"smallest {nv}x{nu}", a, b, u, v, 3, 7, 7, 30, 2.0)
for nv, nu, n in [(600, 800, 300), (768, 1024, 300)]:
    a, b, _ = synth.frame_pair(nv=nv, nu=nu, seed=21, shift=(-4.1, 2.7), zoom=0.003, noise=1.0)
    u, v = synth.keypoints(2000, nv=nv, nu=nu, seed=22)
    report(f"malaga {nv}x{nu}", a, b, u[:n], v[:n], 4, 41, 41, 50, 2.0)
g = np.load("tests/golden/golden.npz")
from tests.test_golden import KLT_CFG
for tag in "slm":
    kw = KLT_CFG[tag]
    report(f"golden {tag}", g["klt_img1"], g["klt_img2"], g["klt_pu"], g["klt_pv"], kw["levels"], kw["wu"], kw["wv"], kw["max_iters"], kw["max_bidir"])
