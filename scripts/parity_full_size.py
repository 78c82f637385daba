"""Parity at BASELINE's full sizes, everything against the oracle (not a sample of it): all 50 000 tracks of a
KITTI-shaped frame pair, and the pose step's 2000 hypotheses x 50 000 points under identical hypotheses.  The test
suite checks the same properties on sizes the oracle finishes in a second on 8 threads (tests/test_gpu_parity.py:
1500 of the 50k tracks, 200 of the hypotheses); this runs the whole thing once on the GPU box's 16 host threads.
    python scripts/parity_full_size.py            (output kept as profiles/round3_parity_full_size.txt)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle
from visual_odometry_amd import ops, synth

K = synth.KITTI_K
THREADS = 16
img1, img2, truth = synth.frame_pair(shift=(3.3, -2.6), zoom=0.004, noise=1.0)
pu, pv = synth.keypoints(50000)
out, valid, score = ops.klt_track(img1, img2, np.stack([pu, pv], 1), 4, 2.0, (41, 41), 50, index_base=0)
qu, qv = out[:, 0], out[:, 1]
t0 = time.perf_counter()
ou, ov, ovalid, oscore, ofb, iters = oracle.klt_track(img1, img2, pu, pv, threads=THREADS)
t_or = time.perf_counter() - t0
flips = valid != ovalid
near = np.abs(ofb - 2.0) < 1e-3
both = valid & ovalid
d = np.hypot(qu - ou, qv - ov)[both]
tu, tv = truth(pu, pv)
print(f"KLT 1241x376, 4 levels, 41x41, <=50 iterations, fwd-bwd 2 px: {len(pu)} keypoints, oracle {t_or:.2f} s on {THREADS} threads")
print(f"  valid on both sides {both.sum()}, validity flips {flips.sum()} (of which within 1e-3 px of the forward-backward threshold: {(flips & near).sum()})")
print(f"  |gpu - oracle| over the {both.sum()} common tracks: max {d.max():.3e} px, 99.9th percentile {np.percentile(d, 99.9):.3e}, median {np.median(d):.3e}   (bar: 0.01 px)")
print(f"  |gpu - truth| median {np.median(np.hypot(qu - tu, qv - tv)[both]):.4f} px; score difference max {np.abs(score - oscore)[both].max():.2e}")
ok = (flips & ~near).sum() == 0 and d.max() < 0.01

xy, X, R_cw, C, is_out = synth.pnp_scene(50000)
for mode in (1, 0):
    t0 = time.perf_counter()
    o = oracle.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj=2.0, seed=4095, mode=mode, min_inliers=4, want_all=True, threads=THREADS)
    t_or = time.perf_counter() - t0
    s = ops.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj_err=2.0, seed=4095, mode=mode, min_inliers=4, hyp_poses=o["poses"], want_all=True)
    same = (np.array_equal(s["counts"], o["counts"]) and np.array_equal(s["costs"], o["costs"]) and np.array_equal(s["inliers"], o["inliers"])
            and (s["best_hyp"], s["best_sol"]) == (o["best_hyp"], o["best_sol"]))
    own = ops.p3p_ransac(xy, X, K, num_hyp=2000, max_reproj_err=2.0, seed=4095, mode=mode, min_inliers=4, want_all=True)
    print(f"P3P-RANSAC {'MSAC' if mode else 'inlier count'}: 2000 hypotheses x 50000 points = {4 * 2000 * 50000 / 1e6:.0f} M pose x point tests, oracle {t_or:.2f} s")
    print(f"  identical hypotheses (the oracle's 8000 poses scored on the GPU): counts, costs, record and inlier set {'bit-equal' if same else 'DIFFER'}; {o['inliers'].sum()} inliers")
    print(f"  own hypotheses (device P3P, libm ulps apart): same record {(own['best_hyp'], own['best_sol']) == (o['best_hyp'], o['best_sol'])}, "
          f"{(own['counts'] == o['counts']).mean() * 100:.2f} % of the per-pose counts equal, inlier sets differ in {(own['inliers'] != o['inliers']).sum()} points")
    ok = ok and same
# ---- optional (--chain N: minutes of oracle time, not part of the test): N frames of the device-resident loop at the
# stress size (25 000 landmarks at the cap + 25 000 seeded candidates, CandidatesToKeep 5000), the state never leaving
# HBM between frames; the oracle's sequential processFrame steps from the state the device held before each frame
if "--chain" in sys.argv:
    from oracle import vo_frame
    from visual_odometry_amd.frame import FrameLoop
    n_chain = int(sys.argv[sys.argv.index("--chain") + 1])
    frames, poses, lift = synth.plane_sequence(n_chain + 1, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    kw = dict(maxNumLandmarks=25000, CandidatesToKeep=5000, MinDistance=1.5, angularThreshold=12.0, maxTemporalRecall=24)
    prev = synth.stress_state(K, lift, vo_frame.new_state, max_age=20)
    loop = FrameLoop(K, frames[0].shape, maxCandidates=32000, **kw)       # room for the new candidates: the reference's arrays grow
    loop.set_state(prev, frames[0])
    for k in range(1, n_chain + 1):
        gpose, gstatus, ginfo = loop.processFrame(frames[k])
        g = loop.get_state()
        t0 = time.perf_counter()
        o, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], prev, K, k, maxNumLandmarks=25000, keep=5000, min_distance=1.5,
                                                       angularThreshold=12.0, maxTemporalRecall=24, threads=THREADS)
        t_or = time.perf_counter() - t0
        same_counts = (ginfo["n_tracked"], ginfo["n_promoted"], ginfo["n_new"], len(g.landmarks), len(g.candidate_keypoints)) == \
                      (info["tracked"], info["promoted"], info["new"], len(o.landmarks), len(o.candidate_keypoints))
        dpose = np.abs(gpose - pose).max()
        dk = np.abs(g.keypoints - o.keypoints).max() if same_counts else float("nan")
        dl = (np.abs(g.landmarks - o.landmarks) / np.maximum(1.0, np.abs(o.landmarks))).max() if same_counts else float("nan")
        same_ids = same_counts and np.array_equal(g.candidate_first_keypoints, o.candidate_first_keypoints) and \
            np.array_equal(g.candidate_time_indxs, o.candidate_time_indxs)
        print(f"frame {k}: {ginfo['n_tracked']} landmarks + {len(prev.candidate_keypoints)} candidates tracked, {ginfo['n_promoted']} promoted "
              f"({info['replaced']} replaced a landmark at the cap), {ginfo['n_new']} new; counts {'equal' if same_counts else 'DIFFER'}, "
              f"candidate identities and ages {'equal' if same_ids else 'DIFFER'}, |pose| {dpose:.1e}, keypoints {dk:.1e} px, landmarks {dl:.1e} rel "
              f"(oracle {t_or:.1f} s)", flush=True)
        ok = ok and same_counts and same_ids and gstatus == status == 0 and dpose < 1e-4 and dk < 0.01 and dl < 1e-4
        prev = g
    loop.close()
print("FULL-SIZE PARITY:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
