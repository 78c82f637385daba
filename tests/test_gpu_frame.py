"""The device-resident frame loop (vo_frame_*) against the oracle's restatement of
VisualOdometry.processFrame (oracle/vo_frame.py), through the C ABI.  Bars as in test_gpu_parity:
tracked positions 0.01 px, landmarks / poses 1e-4 relative; counts and time indices exact."""
import numpy as np
import pytest

from oracle import oracle, vo_frame
from visual_odometry_amd import _lib, synth
from visual_odometry_amd.frame import FrameLoop

pytestmark = pytest.mark.gpu
K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
PX_TOL = 0.01


def start_state(img, lift, n=150):
    u, v, _, _ = oracle.select_candidates(img, np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=n)
    st = vo_frame.new_state()
    st.keypoints = np.stack([u, v], 1).astype(np.float64)
    st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1])
    st.reproError = np.zeros(len(u))
    return st


def assert_states_match(g, o):
    assert len(g.landmarks) == len(o.landmarks) and len(g.candidate_keypoints) == len(o.candidate_keypoints)
    if len(o.landmarks):
        assert np.abs(g.keypoints - o.keypoints).max() < PX_TOL
        np.testing.assert_allclose(g.landmarks, o.landmarks, rtol=1e-4, atol=1e-4)
        assert np.abs(g.reproError - o.reproError).max() < PX_TOL
    if len(o.candidate_keypoints):
        assert np.abs(g.candidate_keypoints - o.candidate_keypoints).max() < PX_TOL
        np.testing.assert_array_equal(g.candidate_first_keypoints, o.candidate_first_keypoints)
        np.testing.assert_allclose(g.candidate_first_poses, o.candidate_first_poses, rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(g.candidate_time_indxs, o.candidate_time_indxs)


@pytest.fixture(scope="module")
def sequence():
    return synth.plane_sequence(14, K)


def test_each_frame_matches_the_oracle_from_the_same_state(ctx, sequence):
    """Same input state on both sides for every frame; the landmark cap (120) is reached, so the
    replace-the-worst policy (:116-122), promotion (:123-127) and ageing (:130-139) all run."""
    frames, poses, lift = sequence
    st = start_state(frames[0], lift)
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
    seen = dict(promoted=0, replaced=0, aged_out=0)
    for k in range(1, len(frames)):
        loop.set_state(st, frames[k - 1], frame_no=k - 1)
        gpose, gstatus, ginfo = loop.processFrame(frames[k])
        st2, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, K, k, maxNumLandmarks=160)
        assert gstatus == status == 0
        np.testing.assert_allclose(gpose, pose, rtol=1e-4, atol=1e-5)
        assert (ginfo["n_tracked"], ginfo["n_promoted"], ginfo["n_new"]) == (info["tracked"], info["promoted"], info["new"])
        assert_states_match(loop.get_state(), st2)
        seen["promoted"] += info["promoted"]; seen["replaced"] += info["replaced"]
        st = st2
    assert seen["promoted"] > 0 and seen["replaced"] > 0
    assert st.candidate_time_indxs.min() < -1          # candidates were carried and aged
    loop.close()


def test_a_chain_of_frames_with_the_state_resident(ctx, sequence):
    """No re-upload: the state stays in HBM for the whole sequence and is only downloaded.  The
    oracle steps from the state the device held before each frame, so a RANSAC tie that resolves
    differently (minimal-sample poses on a plane are that fragile) cannot fork the comparison."""
    frames, poses, lift = sequence
    prev = start_state(frames[0], lift)
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=200, ctx=ctx)
    loop.set_state(prev, frames[0])
    for k in range(1, len(frames)):
        gpose, gstatus, ginfo = loop.processFrame(frames[k])
        g = loop.get_state()
        o, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], prev, K, k, maxNumLandmarks=200)
        assert gstatus == status == 0
        np.testing.assert_allclose(gpose, pose, rtol=1e-4, atol=1e-5)
        assert_states_match(g, o)
        assert loop.counts() == (len(g.landmarks), len(g.candidate_keypoints)) == (ginfo["n_landmarks"], ginfo["n_candidates"])
        prev = g
    # the estimated trajectory follows the true one (minimal-sample poses on a plane: loose bound)
    assert np.abs(gpose[3] - poses[-1][3]).max() < 0.7
    loop.close()


def test_empty_and_tiny_states(ctx, sequence):
    """Fewer than 4 tracked landmarks: status 1 and the identity pose (toolbox semantics), the
    selection step still seeds candidates."""
    frames, _, lift = sequence
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=50, CandidatesToKeep=40, ctx=ctx)
    loop.set_state(vo_frame.new_state(), frames[0])
    pose, status, info = loop.processFrame(frames[1])
    o, opose, ostatus, oinfo = vo_frame.process_frame(frames[0], frames[1], vo_frame.new_state(), K, 1, maxNumLandmarks=50, keep=40)
    assert status == ostatus == 1 and info["n_landmarks"] == 0 and info["n_new"] == oinfo["new"] == 40
    np.testing.assert_array_equal(pose, np.vstack([np.eye(3), np.zeros((1, 3))]))
    assert_states_match(loop.get_state(), o)
    st = start_state(frames[0], lift, n=3)
    loop.set_state(st, frames[0])
    pose, status, info = loop.processFrame(frames[1])
    assert status == 1 and info["n_tracked"] <= 3
    loop.close()


def test_empty_large_state_does_not_wait_for_a_head_that_never_comes(ctx, sequence, monkeypatch):
    """Large states start the candidates' tracking launch behind a HEAD of the landmarks' launch (a count of finished
    workgroups and a flag, polled by a one-lane kernel).  A state without a landmark has no workgroup that could arrive:
    the polling kernel reads the count on the device and returns at once instead of running into its time-out."""
    import ctypes as C
    import time
    frames, _, lift = sequence
    monkeypatch.setenv("VO_FRAME_L_FIRST_MIN", "0")
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=3000, maxCandidates=3000, CandidatesToKeep=40, ctx=ctx)
    monkeypatch.delenv("VO_FRAME_L_FIRST_MIN", raising=False)
    loop.set_state(vo_frame.new_state(), frames[0])
    loop.processFrame(frames[1])                          # buffers, code objects
    loop.set_state(vo_frame.new_state(), frames[0])
    ctx.sync()
    t0 = time.perf_counter()
    pose, status, info = loop.processFrame(frames[1])
    dt = time.perf_counter() - t0
    t = C.c_int(-1)
    ctx.call("vo_dev_head_timed_out", C.byref(t))
    assert status == 1 and info["n_landmarks"] == 0 and info["n_new"] == 40
    assert t.value == 0 and dt < 0.1, (t.value, dt)
    # ... and with landmarks the order gives the frame of the other order
    st = start_state(frames[0], lift, n=140)
    loop.set_state(st, frames[0])
    p1, s1, i1 = loop.processFrame(frames[1])
    g1 = loop.get_state()
    loop.close()
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=3000, maxCandidates=3000, CandidatesToKeep=40, ctx=ctx)
    loop.set_state(st, frames[0])
    p2, s2, i2 = loop.processFrame(frames[1])
    assert s1 == s2 == 0 and np.array_equal(p1, p2) and i1 == i2 and states_equal(g1, loop.get_state())
    loop.close()


def test_candidate_capacity_overflow_is_flagged(ctx, sequence):
    frames, _, lift = sequence
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=200, maxCandidates=130, ctx=ctx)
    loop.set_state(start_state(frames[0], lift), frames[0])
    _, _, i1 = loop.processFrame(frames[1])
    _, _, i2 = loop.processFrame(frames[2])
    assert i1["overflow"] == 0 and i1["n_candidates"] == 100
    assert i2["overflow"] == 1 and i2["n_candidates"] == 130
    loop.close()


def test_bad_calls_raise(ctx, sequence):
    frames, _, lift = sequence
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=20, ctx=ctx)
    with pytest.raises(_lib.VoError):
        loop.processFrame(frames[1])                       # no previous image yet
    with pytest.raises(_lib.VoError):
        loop.set_state(start_state(frames[0], lift, n=60))  # more landmarks than max_landmarks
    with pytest.raises(ValueError):
        loop.set_image(frames[0][:100])
    with pytest.raises(_lib.VoError):
        FrameLoop(K, frames[0].shape, maxNumLandmarks=2, ctx=ctx)
    loop.close()


def test_device_images_and_kitti_shape(ctx):
    """Device-resident uint8 frames at 1241 x 376 with the reference's defaults (300 landmarks)."""
    import torch
    Kk = synth.KITTI_K
    frames, poses, lift = synth.plane_sequence(4, Kk, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    st = start_state(frames[0], lift, n=300)
    dev = [torch.from_numpy(f).to("cuda") for f in frames]
    loop = FrameLoop(Kk, frames[0].shape, ctx=ctx)
    loop.set_state(st, dev[0])
    for k in range(1, 4):
        gpose, gstatus, ginfo = loop.processFrame(dev[k])
        g = loop.get_state()
        st, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, Kk, k)
        assert gstatus == status == 0
        np.testing.assert_allclose(gpose, pose, rtol=1e-4, atol=1e-5)
        assert_states_match(g, st)
        st = g                                   # follow the device's own chain
    loop.close()


def test_merge_beyond_the_lds_copies(ctx, sequence):
    """2600 promotable candidates against a landmark cap of 5000 + 3000 candidates (a pool beyond the one-block
    merge's 4096 LDS entries), then against a cap of 150, where almost every one replaces the worst landmark:
    both through the sort-based merge (list, tile sort, the decisions from the sorted runs, ordered write), the second as one-block
    compactions around it (a state below 2048 entries), the first as one-wave compactions."""
    frames, poses, lift = sequence
    k = 8
    rng = np.random.default_rng(5)
    n_c = 2600
    # candidates seen first in frame 0 and now at their true position in frame k-1
    u0 = rng.uniform(30, 290, n_c); v0 = rng.uniform(30, 210, n_c)
    X = lift(u0, v0)
    R, C = poses[k - 1][:3], poses[k - 1][3]
    uv, z = synth.project(K, R, C, X)
    ok = (uv[:, 0] > 25) & (uv[:, 0] < 295) & (uv[:, 1] > 25) & (uv[:, 1] < 215)
    st = start_state(frames[k - 1], lambda a, b: synth_lift_at(poses[k - 1], a, b), n=140)
    st.candidate_keypoints = uv[ok]
    st.candidate_first_keypoints = np.stack([u0, v0], 1)[ok]
    st.candidate_first_poses = np.repeat(poses[0][None], ok.sum(), 0)
    st.candidate_time_indxs = -np.full(ok.sum(), k - 1, np.int64)
    assert ok.sum() > 2200
    for cap in (5000, 150):
        loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=cap, maxCandidates=3000, ctx=ctx)
        loop.set_state(st, frames[k - 1], frame_no=k - 1)
        gpose, gstatus, ginfo = loop.processFrame(frames[k])
        o, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, K, k, maxNumLandmarks=cap)
        assert gstatus == status == 0 and ginfo["n_promoted"] == info["promoted"] > (2000 if cap == 5000 else 100)
        assert_states_match(loop.get_state(), o)
        loop.close()


def test_merge_at_the_stress_size(ctx):
    """BASELINE configs[3]'s sizes through the device-resident loop: a KITTI-shaped frame, 25 000 landmarks at the cap
    (maxNumLandmarks = 25 000) and ~7 000 candidates of which more than 5 000 are promotable -- the first fill the places
    of the landmarks that failed to track, every later one replaces the worst landmark if its own error is not larger
    (modules/VisualOdometry.m:87-142: max-before-edit :89-90, replace-worst :116-122, append :123-127, ageing :130-139).
    Through the sort-based merge (k_sort_*), against the oracle's sequential walk: same accepted and surviving sets in
    the same order."""
    KK = synth.KITTI_K
    frames, poses, lift = synth.plane_sequence(10, KK, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    k = 8
    rng = np.random.default_rng(11)

    def lift_at(pose, u, v, depth=20.0):
        R, C = pose[:3], pose[3]
        d = R.T @ (np.linalg.inv(KK) @ np.stack([u, v, np.ones_like(u)]))
        lam = (depth - C[2]) / d[2]
        return (C[:, None] + lam * d).T
    n_l, n_c = 25000, 7000
    st = vo_frame.new_state()
    lu = rng.uniform(25, 1241 - 26, n_l); lv = rng.uniform(25, 376 - 26, n_l)
    st.keypoints = np.stack([lu, lv], 1); st.landmarks = lift_at(poses[k - 1], lu, lv); st.reproError = np.zeros(n_l)
    u0 = rng.uniform(30, 1241 - 31, n_c); v0 = rng.uniform(30, 376 - 31, n_c)
    X = lift(u0, v0)
    uv, z = synth.project(KK, poses[k - 1][:3], poses[k - 1][3], X)
    ok = (uv[:, 0] > 25) & (uv[:, 0] < 1241 - 26) & (uv[:, 1] > 25) & (uv[:, 1] < 376 - 26)
    st.candidate_keypoints = uv[ok]; st.candidate_first_keypoints = np.stack([u0, v0], 1)[ok]
    st.candidate_first_poses = np.repeat(poses[0][None], ok.sum(), 0)
    st.candidate_time_indxs = -np.full(ok.sum(), k - 1, np.int64)
    assert ok.sum() > 6000
    loop = FrameLoop(KK, frames[0].shape, maxNumLandmarks=n_l, maxCandidates=8000, CandidatesToKeep=100, ctx=ctx)
    loop.set_state(st, frames[k - 1], frame_no=k - 1)
    gpose, gstatus, ginfo = loop.processFrame(frames[k])
    o, pose, status, info = vo_frame.process_frame(frames[k - 1], frames[k], st, KK, k, maxNumLandmarks=n_l, threads=16)
    assert gstatus == status == 0
    assert ginfo["n_promoted"] == info["promoted"] > 5000 and info["replaced"] > 3000 and ginfo["n_landmarks"] == n_l
    assert_states_match(loop.get_state(), o)
    loop.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_large_states_take_one_wave_compactions_with_the_same_bits(ctx, depth, monkeypatch):
    """States beyond 2048 entries run their ordered compactions as launches of one-wave workgroups that publish their
    counts (compact_wave; k_tracked_w, k_pose_tail_w, k_merge_list_w, k_merge_write_w, k_compact2_w) and -- from 32k
    keypoints a frame, here from 1000 (VO_FRAME_L_FIRST_MIN) -- start the candidates' tracking launch behind the landmarks'.  A chain of frames on a KITTI-shaped sequence -- 3000 landmarks at
    the cap, 3000 seeded candidates of which a few hundred are promoted a frame, new candidates every frame, the
    look-ahead on -- gives the bits of the one-block kernels (VO_FRAME_NARROW=1, read when a loop is created), frame by frame
    and in the final state; and the first frame matches the oracle's restatement of processFrame."""
    KK = synth.KITTI_K
    frames, poses, lift = synth.plane_sequence(8, KK, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    st0 = synth.stress_state(KK, lift, vo_frame.new_state, 3000, 3000, max_age=20)
    kw = dict(maxNumLandmarks=3000, maxCandidates=6000, CandidatesToKeep=400, MinDistance=3.0, angularThreshold=12.0,
              maxTemporalRecall=24)

    def run(narrow):
        if narrow:
            monkeypatch.setenv("VO_FRAME_NARROW", "1")
        else:
            monkeypatch.delenv("VO_FRAME_NARROW", raising=False)
        # the landmarks-first order of the two tracking launches starts at 32k keypoints a frame by itself: here from 1000
        monkeypatch.setenv("VO_FRAME_L_FIRST_MIN", "1000")
        loop = FrameLoop(KK, frames[0].shape, ctx=ctx, **kw)
        loop.set_state(st0, frames[0])
        got, pending = [], 0
        for k in range(1, len(frames)):
            nxt = frames[k + 1] if k + 1 < len(frames) else None
            if depth == 1:
                got.append(loop.processFrame(frames[k], nxt))
            else:
                loop.submit(frames[k], nxt); pending += 1
                if pending == depth:
                    got.append(loop.collect()); pending -= 1
        while pending:
            got.append(loop.collect()); pending -= 1
        final = loop.get_state()
        loop.close()
        return got, final
    wide, wide_final = run(False)
    narrow, narrow_final = run(True)
    monkeypatch.delenv("VO_FRAME_NARROW", raising=False)
    monkeypatch.delenv("VO_FRAME_L_FIRST_MIN", raising=False)
    assert sum(i["n_promoted"] for _, _, i in wide) > 500 and sum(i["n_new"] for _, _, i in wide) > 500
    for (p0, s0, i0), (p1, s1, i1) in zip(wide, narrow):
        assert np.array_equal(p0, p1) and s0 == s1 and i0 == i1
    assert states_equal(wide_final, narrow_final)
    if depth == 1:    # one frame against the oracle, from the same state
        loop = FrameLoop(KK, frames[0].shape, ctx=ctx, **kw)
        loop.set_state(st0, frames[0])
        gpose, gstatus, ginfo = loop.processFrame(frames[1])
        o, pose, status, info = vo_frame.process_frame(frames[0], frames[1], st0, KK, 1, maxNumLandmarks=3000, keep=400,
                                                       min_distance=3.0, angularThreshold=12.0, maxTemporalRecall=24, threads=16)
        assert gstatus == status == 0 and ginfo["n_promoted"] == info["promoted"]
        assert_states_match(loop.get_state(), o)
        loop.close()


def test_a_compaction_wait_that_gives_up_is_reported(ctx, monkeypatch):
    """The one-wave compactions wait for counts that the workgroups in front of them publish; a wait gives up after a
    bounded number of looks so that the grid always drains.  That cannot happen in a correct run -- and if it ever does
    the frame's counts are wrong, which the caller must hear about instead of getting a plausible pose: the wait raises a
    word in pinned host memory and vo_frame_collect returns an error.  Injected here (VO_FRAME_FAULT_INJECT, read when
    a loop is created: the first workgroup of the next tracked-landmark compaction stays silent); afterwards a freshly
    loaded state runs as if nothing had happened -- same bits as a loop that never saw the fault."""
    KK = synth.KITTI_K
    frames, poses, lift = synth.plane_sequence(4, KK, nv=376, nu=1241, depth=20.0, step=(0.2, 0.0, 0.15))
    st0 = synth.stress_state(KK, lift, vo_frame.new_state, 3000, 3000, max_age=20)
    kw = dict(maxNumLandmarks=3000, maxCandidates=6000, CandidatesToKeep=400, MinDistance=3.0, angularThreshold=12.0,
              maxTemporalRecall=24)
    clean = FrameLoop(KK, frames[0].shape, ctx=ctx, **kw)
    clean.set_state(st0, frames[0])
    want = [clean.processFrame(frames[k]) for k in (1, 2)]
    want_final = clean.get_state()
    clean.close()
    monkeypatch.setenv("VO_FRAME_FAULT_INJECT", "1")
    loop = FrameLoop(KK, frames[0].shape, ctx=ctx, **kw)
    monkeypatch.delenv("VO_FRAME_FAULT_INJECT", raising=False)
    loop.set_state(st0, frames[0])
    with pytest.raises(_lib.VoError, match="gave up"):
        loop.processFrame(frames[1])
    loop.set_state(st0, frames[0])                       # the injection was for one launch: the loop is usable again
    got = [loop.processFrame(frames[k]) for k in (1, 2)]
    for (p0, s0, i0), (p1, s1, i1) in zip(got, want):
        assert np.array_equal(p0, p1) and s0 == s1 and i0 == i1
    assert states_equal(loop.get_state(), want_final)
    loop.close()


def synth_lift_at(pose, u, v, depth=12.0):
    """landmarks on the plane Z = depth seen at pixel (u, v) from `pose` ([R_cw; C])"""
    R, C = pose[:3], pose[3]
    d = R.T @ (np.linalg.inv(K) @ np.stack([u, v, np.ones_like(u)]))
    lam = (depth - C[2]) / d[2]
    return (C[:, None] + lam * d).T


def states_equal(a, b):
    return all(np.array_equal(getattr(a, f), getattr(b, f)) for f in
               ("landmarks", "keypoints", "reproError", "candidate_keypoints", "candidate_first_keypoints",
                "candidate_first_poses", "candidate_time_indxs"))


@pytest.mark.parametrize("device_images", [False, True])
def test_look_ahead_changes_nothing_but_the_schedule(ctx, sequence, device_images):
    """vo_frame_announce: every frame's successor announced (its pyramid and corner detection issued beside the
    current frame's tail) against the plain loop: poses, counts and states are the same bits.  Then the cases that
    abandon what was issued: another image than the announced one, a state load, a new previous image."""
    frames, poses, lift = sequence
    if device_images:
        import torch
        imgs = [torch.from_numpy(f).cuda() for f in frames]
    else:
        imgs = [np.ascontiguousarray(f) for f in frames]
    st0 = start_state(frames[0], lift)

    def run(announce):
        loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
        loop.set_state(st0, imgs[0])
        out = []
        for k in range(1, len(imgs)):
            nxt = announce(k)
            p, status, info = loop.processFrame(imgs[k], nxt)
            out.append((p, status, info, loop.get_state() if k % 4 == 0 else None))
        final = loop.get_state()
        loop.close()
        return out, final

    plain, plain_final = run(lambda k: None)
    ahead, ahead_final = run(lambda k: imgs[k + 1] if k + 1 < len(imgs) else None)
    # every third announcement names an image that is not the one processed next; another third announce nothing
    wrong, wrong_final = run(lambda k: (imgs[0] if k % 3 == 0 else imgs[k + 1] if (k % 3 == 1 and k + 1 < len(imgs)) else None))
    for other, other_final in ((ahead, ahead_final), (wrong, wrong_final)):
        for (p0, s0, i0, g0), (p1, s1, i1, g1) in zip(plain, other):
            assert np.array_equal(p0, p1) and s0 == s1 and i0 == i1
            if g0 is not None:
                assert states_equal(g0, g1)
        assert states_equal(plain_final, other_final)
    # another selection on the same ctx between the frames takes the work area the look-ahead used every other
    # frame: that look-ahead is abandoned, the results stay the same
    from visual_odometry_amd import ops
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
    loop.set_state(st0, imgs[0])
    for k in range(1, len(imgs)):
        p, status, info = loop.processFrame(imgs[k], imgs[k + 1] if k + 1 < len(imgs) else None)
        assert np.array_equal(p, plain[k - 1][0]) and info == plain[k - 1][2], k
        ops.select_candidates(frames[0], np.zeros((0, 2), np.float32), keep=20, index_base=0, ctx=ctx)
    assert states_equal(loop.get_state(), plain_final)
    loop.close()
    # a state load and a new previous image between the announcement and its use
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
    loop.set_state(st0, imgs[0])
    loop.processFrame(imgs[1], imgs[2])
    loop.set_state(st0, imgs[0])                       # back to the start: what was issued for imgs[2] is dropped
    p, status, info = loop.processFrame(imgs[1], imgs[2])
    assert np.array_equal(p, plain[0][0]) and info == plain[0][2]
    p, status, info = loop.processFrame(imgs[2])
    assert np.array_equal(p, plain[1][0]) and info == plain[1][2]
    loop.close()


@pytest.mark.parametrize("device_images", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 4])
def test_submitted_frames_equal_processed_frames(ctx, sequence, device_images, depth):
    """vo_frame_submit / vo_frame_collect: the host enqueues up to `depth` frames before it collects the oldest (every
    launch sized by the capacities, the counts read on the device) -- poses, counts and states are the bits of the
    one-call loop, with and without the look-ahead.  Pending frames block the calls that need the host's counts."""
    frames, poses, lift = sequence
    if device_images:
        import torch
        imgs = [torch.from_numpy(f).cuda() for f in frames]
    else:
        imgs = [np.ascontiguousarray(f) for f in frames]
    st0 = start_state(frames[0], lift)
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
    loop.set_state(st0, imgs[0])
    plain = [loop.processFrame(imgs[k]) for k in range(1, len(imgs))]
    plain_final = loop.get_state()
    loop.close()
    for announce in (False, True):
        loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
        loop.set_state(st0, imgs[0])
        got, pending = [], 0
        for k in range(1, len(imgs)):
            loop.submit(imgs[k], imgs[k + 1] if announce and k + 1 < len(imgs) else None)
            pending += 1
            if pending == depth:
                got.append(loop.collect()); pending -= 1
        if depth > 1:
            with pytest.raises(_lib.VoError):
                loop.get_state()                      # frames are pending: the host does not know the counts yet
            with pytest.raises(_lib.VoError):
                loop.processFrame(imgs[1])
        while pending:
            got.append(loop.collect()); pending -= 1
        with pytest.raises(_lib.VoError):
            loop.collect()                            # nothing left
        assert len(got) == len(plain)
        for (p0, s0, i0), (p1, s1, i1) in zip(plain, got):
            assert np.array_equal(p0, p1) and s0 == s1 and i0 == i1
        assert states_equal(loop.get_state(), plain_final)
        loop.close()


def test_more_than_the_allowed_frames_cannot_be_pending(ctx, sequence):
    frames, poses, lift = sequence
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=160, ctx=ctx)
    loop.set_state(start_state(frames[0], lift), frames[0])
    for k in range(1, 1 + FrameLoop.MAX_PENDING):
        loop.submit(frames[k])
    with pytest.raises(_lib.VoError):
        loop.submit(frames[5])
    for k in range(FrameLoop.MAX_PENDING):
        loop.collect()
    loop.close()
