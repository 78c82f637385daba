"""Device-pointer selection entry points (vo_dev_select_candidates and its two-call form vo_dev_select_begin /
vo_dev_select_finish, whose image-only half runs on the ctx's side stream): both against the oracle,
bit-exact, with other work enqueued between the two calls."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from visual_odometry_amd import _lib, synth
from visual_odometry_amd._lib import SelOpts, ptr

pytestmark = pytest.mark.gpu


def _run(ctx, torch, img_d, ex, keep, two_calls, min_distance=7.0, between=None):
    nv, nu = img_d.shape
    o = SelOpts(float(min_distance), 0.001, 5, int(keep), _lib.SCORER_MINEIGEN, 0.08)
    eu = torch.from_numpy(ex[:, 0].astype(np.float32)).cuda()
    ev = torch.from_numpy(ex[:, 1].astype(np.float32)).cuda()
    ou = torch.zeros(max(keep, 1), dtype=torch.float32, device="cuda"); ov = torch.zeros_like(ou)
    lin = torch.zeros(max(keep, 1), dtype=torch.int32, device="cuda")
    met = torch.zeros(max(keep, 1), dtype=torch.float64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    if two_calls:
        ctx.call("vo_dev_select_begin", ptr(img_d), nu, nv, C.byref(o), len(ex) + 5)
        if between:
            between()
        ctx.call("vo_dev_select_finish", 0, ptr(eu), ptr(ev), len(ex), ptr(ou), ptr(ov), ptr(lin), ptr(met), ptr(cnt))
    else:
        ctx.call("vo_dev_select_candidates", ptr(img_d), nu, nv, 0, ptr(eu), ptr(ev), len(ex), C.byref(o), ptr(ou), ptr(ov),
                 ptr(lin), ptr(met), ptr(cnt))
    ctx.sync()
    m = int(cnt.item())
    return ou[:m].cpu().numpy(), ov[:m].cpu().numpy(), lin[:m].cpu().numpy(), met[:m].cpu().numpy()


@pytest.mark.parametrize("shape,n_exist,keep", [((120, 160), 0, 40), ((240, 320), 150, 100), ((376, 1241), 3000, 500)])
def test_two_call_selection_equals_the_single_call_and_the_oracle(ctx, shape, n_exist, keep):
    import torch
    img1, img2, _ = synth.frame_pair(nv=shape[0], nu=shape[1], seed=31, shift=(1.0, 0.5))
    pu, pv = synth.keypoints(max(n_exist, 1), nv=shape[0], nu=shape[1], seed=32, border=5)
    ex = np.stack([pu, pv], 1)[:n_exist]
    img_d = torch.from_numpy(img2).cuda()
    # other work of the same ctx between the two calls: a pyramid build and a KLT launch on the ctx stream
    p = torch.from_numpy(np.stack([pu, pv])).cuda()
    q = torch.zeros_like(p); valid = torch.zeros(p.shape[1], dtype=torch.uint8, device="cuda")
    score = torch.zeros(p.shape[1], dtype=torch.float32, device="cuda")
    img1_d = torch.from_numpy(img1).cuda()

    def between():
        ctx.call("vo_dev_frame_build", 0, ptr(img1_d), shape[1], shape[0], 3)
        ctx.call("vo_dev_frame_build", 1, ptr(img_d), shape[1], shape[0], 3)
        ctx.call("vo_dev_klt_track", 0, 1, ptr(p[0]), ptr(p[1]), p.shape[1], 21, 21, 30, C.c_double(0.01), C.c_double(2.0),
                 ptr(q[0]), ptr(q[1]), ptr(valid), ptr(score), None)
    one = _run(ctx, torch, img_d, ex, keep, False)
    two = _run(ctx, torch, img_d, ex, keep, True, between=between)
    again = _run(ctx, torch, img_d, ex, keep, True)
    ou, ov, olin, omet = oracle.select_candidates(img2, ex[:, 0].astype(np.float32), ex[:, 1].astype(np.float32), min_distance=7,
                                                  keep=keep)
    for got in (one, two, again):
        assert np.array_equal(got[2], olin) and np.array_equal(got[3], omet)
        assert np.array_equal(got[0], ou) and np.array_equal(got[1], ov)


def test_two_call_selection_misuse_is_an_error(ctx):
    import torch
    img = torch.from_numpy(synth.frame_pair(nv=64, nu=80, seed=3)[0]).cuda()
    o = SelOpts(7.0, 0.001, 5, 10, _lib.SCORER_MINEIGEN, 0.08)
    out = torch.zeros(16, dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    e = torch.zeros(8, dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.VoError):                                   # finish without begin
        ctx.call("vo_dev_select_finish", 0, None, None, 0, ptr(out), ptr(out), None, None, ptr(cnt))
    ctx.call("vo_dev_select_begin", ptr(img), 80, 64, C.byref(o), 4)
    ctx.call("vo_dev_select_begin", ptr(img), 80, 64, C.byref(o), 4)    # two may be open
    with pytest.raises(_lib.VoError):                                   # a third may not
        ctx.call("vo_dev_select_begin", ptr(img), 80, 64, C.byref(o), 4)
    with pytest.raises(_lib.VoError):
        ctx.call("vo_dev_frame_prefetch", 0, ptr(img), 80, 64, 3, C.byref(o), 4)
    with pytest.raises(_lib.VoError):                                   # more existing points than announced
        ctx.call("vo_dev_select_finish", 0, ptr(e), ptr(e), 8, ptr(out), ptr(out), None, None, ptr(cnt))
    ctx.call("vo_dev_select_finish", 0, ptr(e), ptr(e), 4, ptr(out), ptr(out), None, None, ptr(cnt))
    ctx.call("vo_dev_select_finish", 0, ptr(e), ptr(e), 4, ptr(out), ptr(out), None, None, ptr(cnt))
    with pytest.raises(_lib.VoError):                                   # the queue is empty again
        ctx.call("vo_dev_select_finish", 0, ptr(e), ptr(e), 4, ptr(out), ptr(out), None, None, ptr(cnt))
    ctx.sync()
    # a single-call selection between begin and finish takes the first work area: that finish says so
    ctx.call("vo_dev_select_begin", ptr(img), 80, 64, C.byref(o), 4)
    ctx.call("vo_dev_select_candidates", ptr(img), 80, 64, 0, ptr(e), ptr(e), 4, C.byref(o), ptr(out), ptr(out), None, None, ptr(cnt))
    with pytest.raises(_lib.VoError):
        ctx.call("vo_dev_select_finish", 0, ptr(e), ptr(e), 4, ptr(out), ptr(out), None, None, ptr(cnt))
    ctx.sync()
    with pytest.raises(_lib.VoError):
        ctx.call("vo_dev_select_begin", None, 80, 64, C.byref(o), 4)


def test_two_open_selections_finish_oldest_first(ctx):
    """Two images begun back to back (the second through vo_dev_frame_prefetch, which also builds its pyramid),
    finished in order: each finish returns its own image's selection, equal to the single call's."""
    import torch
    nv, nu = 150, 200
    imgs = [synth.frame_pair(nv=nv, nu=nu, seed=sd)[0] for sd in (61, 62)]
    pu, pv = synth.keypoints(120, nv=nv, nu=nu, seed=63, border=5)
    ex = np.stack([pu, pv], 1)
    d = [torch.from_numpy(i).cuda() for i in imgs]
    want = [_run(ctx, torch, d[i], ex, 60, False) for i in range(2)]
    assert not np.array_equal(want[0][2], want[1][2])
    o = SelOpts(7.0, 0.001, 5, 60, _lib.SCORER_MINEIGEN, 0.08)
    eu = torch.from_numpy(ex[:, 0].astype(np.float32)).cuda(); ev = torch.from_numpy(ex[:, 1].astype(np.float32)).cuda()
    ctx.call("vo_dev_select_begin", ptr(d[0]), nu, nv, C.byref(o), len(ex))
    ctx.call("vo_dev_frame_prefetch", 2, ptr(d[1]), nu, nv, 3, C.byref(o), len(ex))
    for i in range(2):
        ou = torch.zeros(60, dtype=torch.float32, device="cuda"); ov = torch.zeros_like(ou)
        lin = torch.zeros(60, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.call("vo_dev_select_finish", 0, ptr(eu), ptr(ev), len(ex), ptr(ou), ptr(ov), ptr(lin), None, ptr(cnt))
        ctx.sync()
        m = int(cnt.item())
        assert np.array_equal(lin[:m].cpu().numpy(), want[i][2]) and np.array_equal(ou[:m].cpu().numpy(), want[i][0])
    # the prefetched pyramid serves a tracking call like a built one
    ctx.call("vo_dev_frame_build", 0, ptr(d[0]), nu, nv, 3)
    ctx.call("vo_dev_frame_build", 1, ptr(d[1]), nu, nv, 3)
    p = torch.from_numpy(np.stack([pu, pv])).cuda()
    outs = []
    for sb in (1, 2):
        q = torch.zeros_like(p); valid = torch.zeros(p.shape[1], dtype=torch.uint8, device="cuda")
        score = torch.zeros(p.shape[1], dtype=torch.float32, device="cuda")
        ctx.call("vo_dev_klt_track", 0, sb, ptr(p[0]), ptr(p[1]), p.shape[1], 21, 21, 30, C.c_double(0.01), C.c_double(2.0),
                 ptr(q[0]), ptr(q[1]), ptr(valid), ptr(score), None)
        ctx.sync()
        outs.append((q.cpu().numpy(), valid.cpu().numpy(), score.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


def test_steps_with_work_issued_ahead_equal_serial_steps(ctx):
    """FrameWorkload with the next frame's pyramid and corner detection issued ahead on the side stream
    (vo_dev_frame_prefetch) against the same workload run stage after stage: every step's outputs are the same
    bits, across the wrap of the frame sequence."""
    import torch
    from visual_odometry_amd.device import FrameWorkload
    other = _lib.Context(0)
    try:
        kw = dict(nv=120, nu=160, n_frames=4, num_hyp=64, levels=3, block=21, max_iters=30, keep=40, seed=77)
        serial = FrameWorkload(ctx, 300, 200, ahead=False, **kw)
        ahead = FrameWorkload(other, 300, 200, ahead=True, head=True, **kw)     # one tracking launch, the landmarks its head
        third = _lib.Context(0)
        two = FrameWorkload(third, 300, 200, ahead=True, head=False, **kw)      # two tracking launches on two streams
        for k in range(11):
            serial.step(); ahead.step(); two.step()
            serial.sync(); ahead.sync(); two.sync()
            for name in ("qa", "va", "sa", "rs_out", "rs_inl", "tr_X", "tr_val", "sel_cnt", "sel_u", "sel_v", "sel_lin"):
                for wl in (ahead, two):
                    a, b = getattr(serial, name), getattr(wl, name)
                    if name.startswith("sel_") and name != "sel_cnt":
                        m = int(serial.sel_cnt.item())
                        a, b = a[:m], b[:m]
                    assert torch.equal(a, b) or np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True), (k, name)
            assert int(serial.sel_cnt.item()) > 0
        t = C.c_int(-1)
        other.call("vo_dev_head_timed_out", C.byref(t))
        assert t.value == 0                # no wait for a head ever gave up
        ahead.drain()                      # what was begun ahead is finished; single stages work again
        ahead.select(1); ahead.sync()
        two.drain(); two.sync(); third.close()
    finally:
        other.close()


def test_head_of_a_tracking_launch(ctx):
    """vo_dev_klt_track_head: same results as vo_dev_klt_track for every split of the points (head 0, a few, all), with
    a consumer of the head's results released by vo_dev_wait_head on the lane while the rest of the launch runs; the
    calls that make no sense are refused."""
    import torch
    nv, nu = 120, 160
    img1, img2, _ = synth.frame_pair(nv=nv, nu=nu, seed=91, shift=(1.2, -0.7))
    pu, pv = synth.keypoints(700, nv=nv, nu=nu, seed=92, border=12)
    d1 = torch.from_numpy(img1).cuda(); d2 = torch.from_numpy(img2).cuda()
    p = torch.from_numpy(np.stack([pu, pv])).cuda()
    c = _lib.Context(0)
    try:
        with pytest.raises(_lib.VoError, match="no vo_dev_klt_track_head"):
            c.call("vo_dev_wait_head")
        c.call("vo_dev_frame_build", 0, ptr(d1), nu, nv, 3); c.call("vo_dev_frame_build", 1, ptr(d2), nu, nv, 3)

        def run(n_head):
            q = torch.full_like(p, -1.0); valid = torch.full((p.shape[1],), 7, dtype=torch.uint8, device="cuda")
            copy = torch.full((2, max(n_head or 0, 1)), -2.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            args = (ptr(p[0]), ptr(p[1]))
            tail = (21, 21, 30, C.c_double(0.01), C.c_double(2.0), ptr(q[0]), ptr(q[1]), ptr(valid), None, None)
            if n_head is None:
                c.call("vo_dev_klt_track", 0, 1, *args, p.shape[1], *tail)
            else:
                c.call("vo_ctx_lane_begin"); c.call("vo_ctx_lane_end")
                c.call("vo_dev_klt_track_head", 0, 1, *args, n_head, p.shape[1], *tail)
                with pytest.raises(_lib.VoError):                       # inside the lane the launch is refused
                    c.call("vo_ctx_lane_resume")
                    try:
                        c.call("vo_dev_klt_track_head", 0, 1, *args, n_head, p.shape[1], *tail)
                    finally:
                        c.call("vo_ctx_lane_end")
                if n_head > 0:
                    c.call("vo_ctx_lane_resume")
                    c.call("vo_dev_wait_head")
                    # a consumer on the lane: copies the head's results while the rest of the launch may still run
                    with torch.cuda.stream(torch.cuda.ExternalStream(c.lib.vo_ctx_get_stream(c.h))):
                        copy.copy_(q[:, :n_head])
                    c.call("vo_ctx_lane_end"); c.call("vo_ctx_lane_join")
            c.sync(); torch.cuda.synchronize()
            return q.cpu().numpy(), valid.cpu().numpy(), copy.cpu().numpy()
        want = run(None)
        for n_head in (0, 1, 37, 350, 700):
            q, valid, copy = run(n_head)
            assert np.array_equal(q, want[0], equal_nan=True) and np.array_equal(valid, want[1]), n_head
            if n_head:
                assert np.array_equal(copy, want[0][:, :n_head], equal_nan=True), n_head    # the consumer saw finished results
        t = C.c_int(-1)
        c.call("vo_dev_head_timed_out", C.byref(t))
        assert t.value == 0
    finally:
        c.close()


def test_response_entry_points_leave_open_selections_alone(ctx):
    """vo_corner_response / vo_harris_response / vo_select_keypoints_greedy used to stage through the scratch buffer
    that is also selection work area 0: called between begin and finish they overwrote the detect results (and with a
    larger image freed the block under the pending kernels).  They have a slot of their own now: an open selection
    -- and a frame loop with an announced look-ahead -- gives the same bits with such calls in between."""
    import torch
    from visual_odometry_amd import ops
    from visual_odometry_amd.frame import FrameLoop
    from oracle import vo_frame
    nv, nu = 150, 200
    img = synth.frame_pair(nv=nv, nu=nu, seed=71)[0]
    big = synth.frame_pair(nv=400, nu=640, seed=72)[0]              # larger than anything this ctx's slot has held
    pu, pv = synth.keypoints(100, nv=nv, nu=nu, seed=73, border=5)
    ex = np.stack([pu, pv], 1)
    d = torch.from_numpy(img).cuda()
    want = _run(ctx, torch, d, ex, 50, False)

    def between():
        r1 = ops.corner_response(big, "mineigen", 5, ctx=ctx)
        r2 = ops.corner_response(big, "harris", 9, 0.08, ctx=ctx)
        ops.select_keypoints_greedy(r2, 20, 8, ctx=ctx)
        assert np.array_equal(r1, oracle.corner_response(big, 0, 5))
    got = _run(ctx, torch, d, ex, 50, True, between=between)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # inside a frame loop with the next frame announced (its detect half is in flight on work area 0 or 1)
    K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
    frames, poses, lift = synth.plane_sequence(5, K)
    u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=80)
    st = vo_frame.new_state()
    st.keypoints = np.stack([u, v], 1).astype(np.float64); st.landmarks = lift(u.astype(np.float64), v.astype(np.float64))
    st.reproError = np.zeros(len(u))
    runs = []
    for disturb in (False, True):
        loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=100, ctx=ctx)
        loop.set_state(st, frames[0])
        out = []
        for k in range(1, len(frames)):
            pose, status, info = loop.processFrame(frames[k], frames[k + 1] if k + 1 < len(frames) else None)
            if disturb:
                ops.corner_response(big, "harris", 9, 0.08, ctx=ctx)
            s = loop.get_state()
            out.append((pose, status, s.keypoints, s.candidate_keypoints, s.landmarks))
        loop.close()
        runs.append(out)
    for a, b in zip(*runs):
        assert a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:])) and np.array_equal(a[0], b[0])


def test_lane_priority_is_asked_for_before_the_lane_exists(ctx):
    """The lane has the ctx stream's priority unless vo_ctx_lane_priority asked for more before its first use, and
    the request is only granted to a ctx that is alone on its GPU in this process (tests/test_gpu_a_lane_guard.py
    holds the grant, the refusal and the demotion): here the session's ctx exists, so asking is refused."""
    c = _lib.Context(0)
    try:
        with pytest.raises(_lib.VoError, match="unsupported"):
            c.call("vo_ctx_lane_priority", 1)
        c.call("vo_ctx_lane_priority", 0)                 # giving it up is always allowed until the lane is created
        c.call("vo_ctx_lane_begin"); c.call("vo_ctx_lane_end"); c.call("vo_ctx_lane_join"); c.sync()
        with pytest.raises(_lib.VoError, match="bad argument"):
            c.call("vo_ctx_lane_priority", 0)
    finally:
        c.close()
