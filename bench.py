#!/usr/bin/env python3
"""Headline benchmark: tracked keypoints/s (+ VO frames/s) on KITTI-shaped synthetic frames.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--keypoints 50000] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the hot path of one VO frame with every input resident in HBM (visual_odometry_amd/
device.py): pyramid build, KLT track of the landmark and of the candidate keypoints, P3P-RANSAC pose,
batched triangulation, candidate selection.  Schedule (FrameWorkload.step_ahead): landmarks and candidates are ONE
tracking launch with the landmarks as its head; when the head is out, the pose chain that depends on it starts on the
ctx's lane stream beside the candidates' half (modules/VisualOdometry.m:278-289 need nothing of :71) -- every stream at
normal priority; triangulation and the selection's filter half follow on the main stream, and the next frame's pyramid
and corner detection are issued on the side stream beside the tracking launch.  Default workload = BASELINE.json configs[3] (synthetic
1241x376, 50k keypoints per frame split into 25k landmarks + 25k candidates, 4 levels, 2k RANSAC
hypotheses) -- the configuration the north-star target is quoted on.  N > 1: every rank runs its own
seed-offset sequence (weak scaling, no data-path collective); RCCL gathers the statistics once.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WINDOW_PX = 41 * 41
FLOP_SETUP_PX, FLOP_ITER_PX = 24, 11      # algorithmic flops per window pixel
FP32_PEAK_TFLOPS = 157.3                  # MI355X_MICROARCH.md: fp32 vector peak = fp32-input MFMA peak


def cpu_baseline(args, frames, budget_s=14.0, min_frames=5, max_frames=24):
    """The oracle (CPU restatement, kind 'port') on whole frames of the same workload -- KLT of all keypoints,
    P3P-RANSAC at full size, triangulation, selection -- on consecutive frame pairs: 1 warm-up frame, then at least
    `min_frames` timed frames (more while the budget lasts); the MEDIAN frame time is reported.  Built
    -O3 -march=native on this host for this leg, OpenMP in every stage."""
    import statistics
    import numpy as np
    from oracle import oracle
    from visual_odometry_amd import synth
    build = oracle.use_timing_build()
    host_cpus = os.cpu_count() or 1
    threads = min(16, host_cpus)               # the box's CPU share for one GPU is 16 (os.cpu_count() shows the whole machine)
    nl = args.keypoints // 2
    pu, pv = synth.keypoints(args.keypoints)
    xy, X, *_ = synth.pnp_scene(nl)
    xy0, xy1, poses0, pose1, _ = synth.triangulation_scene(args.keypoints - nl)
    eu, ev = pu.astype("float32"), pv.astype("float32")

    def one_frame(k, n_threads):
        oracle.set_threads(n_threads)
        fa, fb = frames[k % (len(frames) - 1)], frames[k % (len(frames) - 1) + 1]
        t0 = time.perf_counter()
        oracle.klt_track(fa, fb, pu, pv, threads=n_threads)
        t1 = time.perf_counter()
        oracle.p3p_ransac(xy, X, synth.KITTI_K, num_hyp=args.hyp, mode=1, min_inliers=4, threads=n_threads)
        t2 = time.perf_counter()
        oracle.triangulate(xy0, xy1, poses0, pose1, synth.KITTI_K)
        t3 = time.perf_counter()
        oracle.select_candidates(fb, eu, ev, keep=100)
        t4 = time.perf_counter()
        return t4 - t0, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)

    one_frame(0, threads)                                     # warm-up
    times, parts = [], []
    t_all = time.perf_counter()
    while len(times) < max_frames and (len(times) < min_frames or time.perf_counter() - t_all < budget_s):
        t, p = one_frame(len(times) + 1, threads)
        times.append(t); parts.append(p)
    med = statistics.median(times)
    pm = [statistics.median(x) for x in zip(*parts)]
    t1s, _ = one_frame(1, 1)                                  # also single-threaded, one whole frame
    oracle.set_threads(1)
    return {"value": args.keypoints / med, "unit": "keypoints/s", "cores": threads, "kind": "port",
            "frames_per_s": 1.0 / med, "host_cpus": host_cpus, "build": build,
            "single_thread": {"value": args.keypoints / t1s, "unit": "keypoints/s", "frames_per_s": 1.0 / t1s,
                              "sample": f"1 whole frame in {t1s:.1f} s on 1 thread"},
            "sample": f"oracle (C, {build}, OpenMP {threads} threads in every stage): median of {len(times)} whole frames of the bench "
                      f"workload after 1 warm-up frame, {med * 1e3:.0f} ms per frame (KLT {pm[0] * 1e3:.0f} ms, RANSAC {pm[1] * 1e3:.0f} ms, "
                      f"triangulation {pm[2] * 1e3:.0f} ms, selection {pm[3] * 1e3:.0f} ms), nothing scaled"}


def small_workload_leg(ctx, args, FrameWorkload, keypoints=2000, steps=300):
    """Extra leg: the same step at BASELINE configs[1]'s size (KITTI shape, ~2k KLT keypoints), the low end of the
    sweep in profiles/round1_keypoint_sweep.txt; the headline stays the stress size, where the kernels are
    throughput-bound and keypoints/s means something."""
    wl = FrameWorkload(ctx, keypoints // 2, keypoints - keypoints // 2, num_hyp=args.hyp, seed=4099)
    for _ in range(100):
        wl.step()
    wl.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    wl.sync()
    dt = time.perf_counter() - t0
    wl.drain(); wl.sync()
    return {"keypoints_per_frame": keypoints, "hypotheses": args.hyp, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "frames_per_s": steps / dt, "value": keypoints * steps / dt, "unit": "keypoints/s",
            "workload": "the bench step with 2000 keypoints per frame (1000 landmarks + 1000 candidates), everything else as in config"}


def frame_loop_leg(ctx, with_cpu, n_frames=24):
    """Extra leg: the device-resident VisualOdometry.processFrame loop at the
    reference's own sizes (300 landmarks, 100 new candidates per frame, <= 5000 trials at 99 %) on a
    3D-consistent synthetic 1241x376 sequence with the frames resident in HBM; beside it the oracle's
    restatement on the host cores for the first frames of the same sequence."""
    import numpy as np
    import torch
    from visual_odometry_amd import modules, ops, synth
    from visual_odometry_amd.frame import FrameLoop
    K = synth.KITTI_K
    frames, poses, lift = synth.plane_sequence(n_frames, K, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    kp = ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), min_distance=7, min_quality=0.001, keep=300,
                               index_base=0, ctx=ctx).astype(np.float64)
    st = modules._new_state()
    st.keypoints = kp; st.landmarks = lift(kp[:, 0], kp[:, 1])
    st.reproError = np.zeros(len(kp))
    st.candidate_first_poses = np.zeros((0, 4, 3))
    dev = [torch.from_numpy(f).to(f"cuda:{ctx.device}") for f in frames]
    loop = FrameLoop(K, frames[0].shape, ctx=ctx)

    def passes(lp, state, look_ahead, depth=1):
        """best of 10 passes over the sequence (pass 0 warms the scratch buffers up, the others the clocks).
        look_ahead: every frame's successor is announced (vo_frame_announce), as a loop fed by the decode pool does.
        depth > 1: vo_frame_submit / vo_frame_collect with up to `depth` frames enqueued before the oldest is
        collected (the host does not wait for a frame before it enqueues the next); every frame is collected inside
        the timed region."""
        times = []
        for _ in range(10):
            lp.set_state(state, dev[0])
            ctx.sync()
            t0 = time.perf_counter()
            if depth == 1:
                for k in range(1, n_frames):
                    res = lp.processFrame(dev[k], dev[k + 1] if look_ahead and k + 1 < n_frames else None)
            else:
                pending = 0
                for k in range(1, n_frames):
                    lp.submit(dev[k], dev[k + 1] if look_ahead and k + 1 < n_frames else None)
                    pending += 1
                    if pending == depth:
                        res = lp.collect(); pending -= 1
                while pending:
                    res = lp.collect(); pending -= 1
            times.append(time.perf_counter() - t0)
        import statistics
        return min(times), statistics.median(times[1:]), res   # best pass, median of passes 2-10 (pass 1 warms the buffers up)

    plain, _, _ = passes(loop, st, False)
    sync_best, _, _ = passes(loop, st, True)
    best, med, (pose, status, info) = passes(loop, st, True, depth=2)
    out = {"frames": n_frames - 1, "ms_per_frame": 1e3 * best / (n_frames - 1), "frames_per_s": (n_frames - 1) / best,
           "ms_per_frame_median": 1e3 * med / (n_frames - 1),   # ms_per_frame is the best of 10 passes, this the median of passes 2-10
           "ms_per_frame_one_call_per_frame": 1e3 * sync_best / (n_frames - 1),
           "ms_per_frame_without_look_ahead": 1e3 * plain / (n_frames - 1),
           "final_landmarks": info["n_landmarks"], "final_candidates": info["n_candidates"], "pose_status": status,
           "workload": "vo_frame_process: 1241x376, maxNumLandmarks 300, 100 new candidates/frame, 4-level 41x41 KLT, "
                       "<= 5000 P3P trials (99 %), min-eigen selection; frames resident in HBM, state never leaves HBM; "
                       "each frame's successor announced (its pyramid and corner detection run beside the frame's tail); "
                       "ms_per_frame: vo_frame_submit / vo_frame_collect with two frames in flight (the host enqueues frame "
                       "k+1 before it collects frame k), ms_per_frame_one_call_per_frame: vo_frame_process"}
    loop.close()
    # the same loop at BASELINE configs[1]'s size: ~2k tracked keypoints per KITTI-shaped frame
    kp2 = ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), min_distance=7, min_quality=0.001, keep=2000,
                                index_base=0, ctx=ctx).astype(np.float64)
    st2 = modules._new_state()
    st2.keypoints = kp2; st2.landmarks = lift(kp2[:, 0], kp2[:, 1]); st2.reproError = np.zeros(len(kp2))
    st2.candidate_first_poses = np.zeros((0, 4, 3))
    loop2 = FrameLoop(K, frames[0].shape, maxNumLandmarks=2000, CandidatesToKeep=300, ctx=ctx)
    plain2, _, _ = passes(loop2, st2, False)
    sync2, _, _ = passes(loop2, st2, True)
    best2, med2, (pose2, status2, info2) = passes(loop2, st2, True, depth=2)
    loop2.close()
    out["at_2k_landmarks"] = {"ms_per_frame": 1e3 * best2 / (n_frames - 1), "frames_per_s": (n_frames - 1) / best2,
                              "ms_per_frame_median": 1e3 * med2 / (n_frames - 1),
                              "ms_per_frame_one_call_per_frame": 1e3 * sync2 / (n_frames - 1),
                              "ms_per_frame_without_look_ahead": 1e3 * plain2 / (n_frames - 1),
                              "start_landmarks": len(kp2), "final_landmarks": info2["n_landmarks"],
                              "final_candidates": info2["n_candidates"], "tracked_last_frame": info2["n_tracked"],
                              "pose_status": status2,
                              "workload": "as above with maxNumLandmarks 2000 and 300 new candidates per frame "
                                          "(BASELINE configs[1]: ~2k KLT keypoints on a KITTI-shaped frame)"}
    # The loop at the stress size (BASELINE configs[3] / SURVEY 8d stress variant): 25 000 landmarks at the cap, room for
    # 25 000 candidates, up to 5 000 new candidates a frame -- what "50k keypoints per frame" means for the state chain of
    # processFrame rather than for five stages on canned inputs.  MinDistance is 1.5 px here: at the reference's 7 px a
    # KITTI frame has no room for 50k points (discs of radius 7 around them cover it 16 times over), so no new candidate
    # would ever pass the filter.  The candidates are seeded: first seen 1 .. 20 frames before frame 0 (poses extrapolated
    # backwards along the track, first keypoints = their projections there); with the angular threshold at 12 degrees
    # (0.57 degrees of parallax a frame at this depth) a cohort of 1 250 becomes promotable every frame, each promotion
    # replacing the worst landmark at the cap, and 500 - 1 800 new candidates are appended a frame: while the candidate
    # arrays are full (maxCandidates = the seeded 25 000; the first eight frames, `overflow` set in their records) as many
    # as promotions and expiries make room for, afterwards what the scene feeds (a 1241 x 376 frame holds ~13k corners of
    # this detector, 500 - 800 of them new a frame: the seeded 50k settle towards that, see keypoints_mean;
    # scripts/stress_counts.py prints the counts frame by frame, with and without room).
    n_l = n_c = 25000
    st5 = synth.stress_state(K, lift, modules._new_state, n_l, n_c, max_age=20)
    loop5 = FrameLoop(K, frames[0].shape, maxNumLandmarks=n_l, maxCandidates=n_c, CandidatesToKeep=5000, MinDistance=1.5,
                      angularThreshold=12.0, maxTemporalRecall=24, ctx=ctx)
    best5, med5, (pose5, status5, info5) = passes(loop5, st5, True, depth=2)
    loop5.set_state(st5, dev[0])                           # one more pass, a call per frame, for the counts frame by frame
    held = []
    for k in range(1, n_frames):
        _, _, inf = loop5.processFrame(dev[k], dev[k + 1] if k + 1 < n_frames else None)
        held.append((inf["n_landmarks"], inf["n_candidates"], inf["n_promoted"], inf["n_new"]))
    loop5.close()
    held = np.array(held, dtype=np.float64)
    # keypoints a frame tracks = the state it starts from: landmarks + candidates after the frame before (50 000 for frame 1)
    tracked = np.concatenate([[n_l + n_c], held[:-1, 0] + held[:-1, 1]])
    out["at_50k_keypoints"] = {"ms_per_frame": 1e3 * best5 / (n_frames - 1), "ms_per_frame_median": 1e3 * med5 / (n_frames - 1),
                               "frames_per_s": (n_frames - 1) / med5, "keypoints_mean": float(tracked.mean()),
                               "keypoints_first_frame": int(tracked[0]), "keypoints_last_frame": int(tracked[-1]),
                               "promoted_mean": float(held[:, 2].mean()), "new_mean": float(held[:, 3].mean()),
                               "final_landmarks": info5["n_landmarks"], "final_candidates": info5["n_candidates"],
                               "pose_status": status5,
                               "workload": "vo_frame_submit / _collect, two frames in flight: maxNumLandmarks 25000 (the state starts "
                                           "at the cap) + 25000 candidates of ages 1..20 frames (maxCandidates 25000), angular "
                                           "threshold 12 degrees and maxTemporalRecall 24 (a cohort of 1250 candidates becomes "
                                           "promotable a frame), CandidatesToKeep 5000, MinDistance 1.5 px (at 7 px a KITTI frame has "
                                           "no room for 50k points), everything else as above; the merge is the sort-based form, the "
                                           "compactions the one-wave form; keypoints_mean = keypoints tracked per frame over the "
                                           "pass (new candidates: as many as there is room for while the candidate arrays are full -- the first eight frames --, then "
                                           "what the scene feeds, fewer than expire); frames_per_s from the median pass"}
    if with_cpu:                                          # cpu_baseline of this leg: the oracle's restatement
        from oracle import vo_frame
        threads = min(16, os.cpu_count() or 1)
        m = min(6, n_frames - 1)
        t0 = time.perf_counter()
        for k in range(1, m + 1):
            st, *_ = vo_frame.process_frame(frames[k - 1], frames[k], st, K, k, threads=threads)
        out["cpu_ms_per_frame"] = 1e3 * (time.perf_counter() - t0) / m
        out["cpu_cores"] = threads
    out["ingest"] = ingest_leg(ctx, frames, loop_state=st if not with_cpu else None)
    return out


def ingest_leg(ctx, frames, loop_state=None, n_files=96, threads=8):
    """Extra leg: the decode pool on KITTI-shaped gray PNG files (written here with zlib, rows
    filtered with 'Up'), alone: files -> device images per second on `threads` host threads."""
    import shutil
    import struct
    import tempfile
    import zlib
    import numpy as np
    from visual_odometry_amd import ingest

    def png(img):
        h, w = img.shape
        d = img.astype(np.int16)
        d[1:] -= img[:-1].astype(np.int16)               # filter type 2 (Up)
        raw = np.concatenate([np.full((h, 1), 2, np.uint8), (d & 255).astype(np.uint8)], 1).tobytes()

        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    tmp = tempfile.mkdtemp(prefix="vo_bench_ingest_")
    try:
        blobs = [png(f) for f in frames[:8]]
        paths = []
        for i in range(n_files):
            p = os.path.join(tmp, "%06d.png" % i)
            with open(p, "wb") as f:
                f.write(blobs[i % len(blobs)])
            paths.append(p)
        best = None
        for _ in range(2):
            fs = ingest.FrameStream(paths, threads=threads, ctx=ctx)
            ctx.sync()
            t0 = time.perf_counter()
            n = sum(1 for _ in fs)
            ctx.sync()
            dt = time.perf_counter() - t0
            st = fs.stats()
            fs.close()
            best = dt if best is None else min(best, dt)
        return {"files": n, "frames_per_s": n / best, "threads": st.threads, "host_ms_per_file": 1e3 * st.decode_seconds / n,
                "workload": "vo_ingest_*: %d gray PNG files of 1241x376 (%d kB each, 'Up' rows) -> device images, decode pool "
                            "alone" % (n, len(blobs[0]) // 1000)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--keypoints", type=int, default=50000, help="keypoints tracked per frame (landmarks + candidates)")
    ap.add_argument("--hyp", type=int, default=2000)
    ap.add_argument("--spin-up", type=float, default=0.5, help="seconds of untimed steps before the warm-up (clock ramp)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-frame-loop", action="store_true", help="skip the device-resident processFrame leg")
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a launcher: this process starts the N ranks itself (before torch is imported
    # or any GPU call is made -- the parent never holds a HIP context) and relays their verdict.
    from visual_odometry_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    if os.environ.get("VO_BENCH_RENDEZVOUS_ONLY"):
        # CPU rehearsal of the N-rank start (tests/test_dist_cpu.py): rendezvous over gloo, the statistics
        # collective, one line from rank 0 -- and nothing that needs a GPU
        from visual_odometry_amd import dist
        rank, local_rank, world = dist.init("gloo")
        if world != args.gpus:
            sys.exit(2)
        dist.barrier()
        stats = dist.gather_stats({"frames": 1, "keypoints_tracked": args.keypoints, "elapsed_s": 1.0 + rank})
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": world, "frames": stats["frames"], "elapsed_s": stats["elapsed_s"],
                              "per_rank_ms_per_step": dist.per_rank_ms_per_step(stats, 1)}))
        dist.shutdown()
        return

    import torch
    from visual_odometry_amd import _lib, dist
    from visual_odometry_amd.device import FrameWorkload

    # VO_REHEARSE_ONE_GPU=1: every rank on device 0 with gloo for the statistics -- a rehearsal of the
    # N > 1 code path on a one-GPU box (the numbers mean nothing); never set by the driver
    rehearse = bool(os.environ.get("VO_REHEARSE_ONE_GPU"))
    if rehearse:
        # several processes on ONE GPU: a high-priority lane stream of one process starved the other's streams for good
        # (a hang, found by this rehearsal in round 2; the library now refuses the priority there) -- never asked for here
        os.environ.setdefault("VO_LANE_NORMAL_PRIORITY", "1")
    rank, local_rank, world = dist.init("gloo" if rehearse else None)
    if rehearse:
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no HIP device (there is no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    # the communicator is set up by its first collective (seconds, with the GPU idle): here, in front of the spin-up, and
    # not inside the barrier that opens the timed region -- the steps behind that one would start on a device whose
    # clocks have come down (the first few hundred milliseconds after an idle period run ~3 % slower, below)
    dist.barrier()
    ctx = _lib.Context(local_rank)
    nl = args.keypoints // 2
    nc = args.keypoints - nl
    # default configuration: every stream at normal priority (vo_ctx_lane_priority is not asked for; VO_LANE_PRIORITY=1
    # asks for it, an A/B switch).  The pose chain gets the landmark tracks early by ORDER: they are the head of the
    # one tracking launch (vo_dev_klt_track_head), and the lane's chain starts when the head is out.
    wl = FrameWorkload(ctx, nl, nc, num_hyp=args.hyp, seed=1023 + 7919 * rank,
                       lane_priority=bool(os.environ.get("VO_LANE_PRIORITY")) and not rehearse)

    # Device spin-up, untimed and before the W warm-up steps: the first few hundred milliseconds after an idle
    # period run ~3 % slower than the steady state (clocks / power state: 1.26-1.27 ms per step right after the
    # start, 1.23 ms from ~0.3 s on, measured on the pool's boxes); a sustained frame loop lives in the steady
    # state, so that is what the K timed steps should see.  Reported as "spin_up_s" in the line.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_up:
        for _ in range(20):
            wl.step()
        wl.sync()
    for _ in range(args.warmup):
        wl.step()
    wl.sync(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    wl.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dist.barrier()
    # the pose chain of a step starts behind a bounded device-side wait for the head of the tracking launch: had it ever
    # given up, the steps timed above would not be the workload -- say so and fail instead of printing a number
    head_gave_up = C.c_int(0)
    ctx.call("vo_dev_head_timed_out", C.byref(head_gave_up))
    if head_gave_up.value:
        print("bench.py: a wait for the head of a tracking launch gave up during the timed steps; no result", file=sys.stderr)
        sys.exit(3)
    valid = int(wl.vl.sum().item()) + int(wl.vc.sum().item())
    inl = int(wl.rs_inl.sum().item())
    stats = dist.gather_stats({"frames": args.steps, "keypoints_tracked": args.steps * args.keypoints,
                               "keypoints_valid": valid, "ransac_inliers": inl, "elapsed_s": elapsed})

    # ---- dominant kernel: k_lk, timed in situ.  Further whole steps are run with HIP events (on the
    # ctx stream, where the kernel is launched) around the frame's KLT launch, so the launch sees the
    # same clocks and cache state as in the timed region and as under rocprofv3; algorithmic flops
    # from the executed iteration / set-up counters of exactly those launches.
    roof = None
    stage_ms = {}
    if rank == 0:
        reps = max(1, min(args.steps, 30))
        wl.sync()
        ctx.counters(reset=True)
        respin = 60
        for _ in range(respin): # untimed: the statistics collective above left the device idle for a moment and the
            wl.step()           # clocks take ~25 steps to come back (the kernel runs 1.1 ms right after, 0.86 steady)
        for _ in range(reps):   # whole steps with a lap around the KLT launch (nothing runs beside it: the work
            wl.step(lap=True)   # issued ahead starts behind the tracking launch, see FrameWorkload.step_ahead)
        ms, nlaps = ctx.laps_ms()                         # one synchronisation for all laps
        ms /= nlaps
        wl.sync()
        it, kp, su, _ = ctx.counters(reset=True)
        flops = WINDOW_PX * (FLOP_ITER_PX * it + FLOP_SETUP_PX * su) / (respin + reps)   # the counters saw every launch since the reset
        ach = flops / (ms * 1e-3) / 1e12
        # HBM-side bytes per launch and the VALU issue utilisation come from the rocprofv3 --pmc passes over this same
        # kernel and workload (scripts/profile_round.sh -> profiles/round3_k_lk_pmc.json); they are quoted only while the
        # kernel's sources are the ones that were profiled (sha256 in the file)
        traffic = valu_util = None
        pmc_note = "no PMC summary for this build of k_lk: traffic and valu_util not quoted"
        try:
            import hashlib
            pmc = json.load(open(os.path.join(ROOT, "profiles", "round3_k_lk_pmc.json")))
            sha = hashlib.sha256()
            for f in ("vo_klt.hip", "vo_image.hip", "vo_common.h"):
                sha.update(open(os.path.join(ROOT, "visual_odometry_amd", "csrc", f), "rb").read())
            if pmc.get("keypoints_per_launch") == args.keypoints and pmc.get("source_sha256") == sha.hexdigest():
                traffic = pmc["hbm_bytes_per_launch"]
                valu_util = pmc.get("valu_issue_util")
                pmc_note = ("traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (fabric-side, Infinity-Cache hits included; "
                            "FETCH_SIZE doubled for the 16-B/lane loads) and valu_util = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x "
                            "GRBM_GUI_ACTIVE / 8) from profiles/round3_k_lk_pmc.json, PMC passes of this same kernel source; by the same "
                            "formula a kernel of nothing but v_fma_f32 reaches 0.67 at this kernel's 4 waves per SIMD and 0.75-0.82 "
                            "at 8 (profiles/round2_valu_clock_pmc.txt)")
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "valu", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                "traffic": traffic, "valu_util": valu_util, "kernel": "k_lk<41x41>", "launch_ms": ms,
                "keypoints_per_launch": args.keypoints,
                "launches_timed": reps, "iters_per_keypoint": it / max(kp, 1), "setups_per_keypoint": su / max(kp, 1),
                "note": "fp32 VALU-bound kernel (no MFMA: nothing here is a dense contraction; the schema's bound names are "
                        "hbm / mfma, this one is the vector ALU); peak = fp32 vector rate at 2.4 GHz (the kernel holds ~2.05 GHz); "
                        "flops = 1681 px x (11/iteration + 24/set-up), SURVEY 8d: ALGORITHMIC flops, the correlation-cell "
                        "formulation executes fewer; launch_ms = HIP events around ONE 50k-keypoint launch (landmarks + candidates, "
                        "nothing beside it) inside whole frame steps, read back once at the end -- the timed steps above run the "
                        "same launch with the pose chain beside its second half (profiles/round3_k_lk_by_grid.txt has both); " + pmc_note}
        wl.drain()              # single stages from here on: finish what was begun ahead
        a, b = wl.k % 2, (wl.k + 1) % 2
        for name, fn in (("pyramid", lambda: wl.build(b, (wl.k % wl.n_frames) + 1)), ("klt_landmarks+candidates", lambda: wl.track_all(a, b)),
                         ("ransac", wl.pose), ("triangulate", wl.triangulate), ("select", lambda: wl.select(1))):
            fn(); wl.sync()
            ctx.timer_start()
            for _ in range(5):
                fn()
            stage_ms[name] = ctx.timer_stop_ms() / 5

        # ---- the pose chain alone: its scoring kernel against the fp64 vector peak, and the chain's time at other
        # outlier ratios (the sieve leaves out what cannot beat the first stage's record: how much that is depends on
        # the scene; the timed steps score the scene of SURVEY 8d, 20 % outliers)
        pose_chain = None
        try:
            import numpy as np
            from visual_odometry_amd import synth
            ctx.call("vo_ctx_time_score", 1)
            t = C.c_uint64()
            ctx.call("vo_ctx_read_score_tests", C.byref(t), 1)
            wl.pose(); wl.sync(); ctx.laps_ms(); ctx.call("vo_ctx_read_score_tests", C.byref(t), 1)     # warm, then clear
            runs = 10
            for _ in range(runs):
                wl.pose()
            ms_score, nlaps = ctx.laps_ms()
            ctx.call("vo_ctx_read_score_tests", C.byref(t), 1)
            ctx.call("vo_ctx_time_score", 0)
            flop = 26.0 * t.value                         # 11 fma + 3 mul + 1 compare per pose x point test (vo_dev_math.h)
            ach64 = flop / (ms_score * 1e-3) / 1e12
            by_outliers = {}
            keep_xy, keep_X = wl.rs_xy, wl.rs_X
            for frac in (0.2, 0.5, 0.75):
                xy, X, *_ = synth.pnp_scene(wl.nl, seed=1023 + 3072, nv=wl.nv, nu=wl.nu, outlier_frac=frac)
                wl.rs_xy = torch.from_numpy(xy).to(wl.dev); wl.rs_X = torch.from_numpy(X).to(wl.dev)
                wl.pose(); wl.sync()
                ctx.timer_start()
                for _ in range(5):
                    wl.pose()
                by_outliers["%.2f" % frac] = ctx.timer_stop_ms() / 5
            wl.rs_xy, wl.rs_X = keep_xy, keep_X
            pose_chain = {"k_score": {"bound": "valu_fp64", "achieved": ach64, "peak": 78.6, "unit": "TFLOP/s", "frac": ach64 / 78.6,
                                      "tests_per_run": t.value / runs, "launches_per_run": nlaps / runs, "ms_per_run": ms_score / runs,
                                      "note": "fp64 flop actually executed by the pose step's scoring kernel (26 per pose x point "
                                              "test: 11 fma, 3 mul, 1 compare; inliers also pay a division) / the summed duration of "
                                              "its launches (HIP events around each, the chain alone on the chip) / the fp64 vector "
                                              "peak; the sieve scored this many tests instead of 4 x hypotheses x points = %d"
                                              % (4 * args.hyp * wl.nl)},
                          "ms_alone_by_outlier_fraction": by_outliers,
                          "workload": f"{args.hyp} hypotheses x {wl.nl} points, MSAC, every hypothesis counts (no confidence)"}
        except Exception as e:                            # a measurement leg must never cost the headline
            pose_chain = {"error": repr(e)}

    if rank == 0:
        value = stats["keypoints_tracked"] / stats["elapsed_s"]
        line = {"metric": "tracked_keypoints_per_s", "value": value, "unit": "keypoints/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "spin_up_s": args.spin_up, "ms_per_step": 1e3 * stats["elapsed_s"] / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "frames_per_s": stats["frames"] / stats["elapsed_s"],
                "config": {"workload": "BASELINE configs[3]: synthetic 1241x376 frames, 16-frame loop per GPU, "
                                       f"{args.keypoints} keypoints/frame ({nl} landmarks + {nc} candidates), 4-level pyramid, "
                                       f"41x41 window, <=50 iterations, fwd-bwd 2 px, {args.hyp} P3P-RANSAC hypotheses (MSAC) on the "
                                       "landmarks, batched triangulation of the candidates, min-eigen selection (keep 100)",
                           "schedule": "one tracking launch, landmarks first; when that head is out the pose chain starts on a "
                                       "second stream of the same (normal) priority beside the candidates' half; triangulation "
                                       "and selection filter behind both; next frame's pyramid and corner detection issued ahead "
                                       "beside the tracking launch (VO_NO_HEAD=1 / VO_NO_LANE=1 / VO_NO_PREFETCH=1 switch them off)",
                           "lane_priority": bool(os.environ.get("VO_LANE_PRIORITY")) and not rehearse,
                           "keypoints_per_frame": args.keypoints, "hypotheses": args.hyp,
                           "parallelism": f"{world} independent sequence(s), one per GPU"},
                # every rank's own time per step (min / max / each, rank order): ms_per_step above is the slowest rank's
                "per_rank_ms_per_step": dist.per_rank_ms_per_step(stats, args.steps),
                "valid_fraction": stats["keypoints_valid"] / (world * args.keypoints),
                "stage_ms": stage_ms, "roofline": roof, "pose_chain": pose_chain}
        if not args.no_cpu and world == 1:                # the CPU side-by-side belongs to the one-GPU line only
            frames = [f.cpu().numpy() for f in wl.frames[:5]]
            line["cpu_baseline"] = cpu_baseline(args, frames)
        if not args.no_frame_loop and world == 1:         # extra legs: one GPU, while no other rank waits
            line["at_2k_keypoints"] = small_workload_leg(ctx, args, FrameWorkload)
            line["frame_loop"] = frame_loop_leg(ctx, not args.no_cpu)
        print(json.dumps(line))
    dist.barrier()
    dist.shutdown()


if __name__ == "__main__":
    main()
