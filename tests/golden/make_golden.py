"""Emit tests/golden/golden.npz: small seeded inputs and the ORACLE's outputs for them.

The reference (pure MATLAB over closed toolboxes, no datasets, no numeric tests) pins no result, and
nothing of it can run here, so these vectors come from the CPU restatement under
oracle/ -- they freeze the normative definitions N1-N9 of oracle/vo_oracle.h so that neither the
oracle nor the HIP path can drift silently.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402

K = synth.KITTI_K
out = {}

# ---- KLT: 96 x 128 pair, points down to the border (windows hang outside at coarse levels)
img1, img2, _ = synth.frame_pair(nv=96, nu=128, seed=21, shift=(1.7, -1.2), zoom=0.003, noise=0.5)
r = np.random.Generator(np.random.Philox(22))
pu = r.uniform(2, 125, 48).astype(np.float32).astype(np.float64)
pv = r.uniform(2, 93, 48).astype(np.float32).astype(np.float64)
out.update(klt_img1=img1, klt_img2=img2, klt_pu=pu, klt_pv=pv)
for tag, kw in (("s", dict(levels=3, wu=21, wv=21, max_iters=30, max_bidir=1.0)),
                ("l", dict(levels=4, wu=41, wv=41, max_iters=50, max_bidir=2.0)),
                ("m", dict(levels=2, wu=31, wv=25, max_iters=30, max_bidir=-1.0))):
    qu, qv, valid, score, fb, iters = oracle.klt_track(img1, img2, pu, pv, **kw)
    out.update({f"klt_{tag}_qu": qu, f"klt_{tag}_qv": qv, f"klt_{tag}_valid": valid, f"klt_{tag}_score": score,
                f"klt_{tag}_iters": iters, f"klt_{tag}_fb": fb})

# ---- selection
sel_img, _, _ = synth.frame_pair(nv=96, nu=128, seed=23)
eu = r.uniform(0, 127, 40).astype(np.float32); ev = r.uniform(0, 95, 40).astype(np.float32)
out.update(sel_img=sel_img, sel_eu=eu, sel_ev=ev)
for tag, kw in (("me_rm", dict(scorer=0, filter_size=5, col_major=False)),
                ("me_cm", dict(scorer=0, filter_size=5, col_major=True)),
                ("ha_rm", dict(scorer=1, filter_size=3, kappa=0.08, col_major=False))):
    u, v, lin, met = oracle.select_candidates(sel_img, eu, ev, min_distance=7.0, min_quality=0.001, keep=25, **kw)
    out.update({f"sel_{tag}_u": u, f"sel_{tag}_v": v, f"sel_{tag}_lin": lin, f"sel_{tag}_metric": met})
h_img = r.integers(0, 256, (32, 40)).astype(np.uint8)
resp = oracle.corner_response(h_img, scorer=1, filter_size=9, kappa=0.08)
gu, gv = oracle.select_keypoints_greedy(resp, 10, 3)
out.update(harris_img=h_img, harris_resp=resp, greedy_u=gu, greedy_v=gv,
           mineig_resp=oracle.corner_response(h_img, scorer=0, filter_size=5))

# ---- geometry
xy0, xy1, poses0, pose1, X = synth.triangulation_scene(64)
Xe, rep, ang, valid = oracle.triangulate(xy0, xy1, poses0, pose1, K)
out.update(tri_xy0=xy0, tri_xy1=xy1, tri_poses0=poses0, tri_pose1=pose1, tri_X=Xe, tri_reproj=rep, tri_angle=ang,
           tri_valid=valid)

# ---- P3P / RANSAC
P = r.uniform(-3, 3, (32, 3, 3)) + np.array([0, 0, 8.0])
f = P + r.normal(0, 0.5, P.shape); f /= np.linalg.norm(f, axis=2, keepdims=True)
out.update(p3p_P=P, p3p_f=f, p3p_poses=np.stack([oracle.p3p(P[i], f[i]) for i in range(32)]))
xy, Xw, *_ = synth.pnp_scene(200, seed=31)
out.update(rs_xy=xy, rs_X=Xw)
for mode in (0, 1):
    g = oracle.p3p_ransac(xy, Xw, K, num_hyp=64, max_reproj=2.0, seed=4095, mode=mode, min_inliers=4, want_all=True)
    out.update({f"rs{mode}_counts": g["counts"], f"rs{mode}_costs": g["costs"], f"rs{mode}_inliers": g["inliers"],
                f"rs{mode}_best": np.array([g["best_hyp"], g["best_sol"], g["trials"], g["status"]]),
                f"rs{mode}_R": g["R"], f"rs{mode}_t": g["t"], f"rs{mode}_poses": g["poses"]})

path = os.path.join(ROOT, "tests", "golden", "golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
