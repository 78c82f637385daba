"""The high-priority lane is closed off at the API (include/vo_hip.h, vo_ctx_lane_priority).

What was seen in round 2 (gpurun_out/t_dev.log, DESIGN "Schedule of a step"): ONE process, a second vo_ctx created
and destroyed, then the first ctx's frame loop creating its lane as a high-priority stream -- vo_frame_process never
returned.  The library now keeps a per-device record of every ctx and decode pool the process has held:
  * the request is granted only to a ctx that is, and always was, alone on its GPU (VO_ERR_UNSUPPORTED otherwise:
    the lane then simply has the ctx stream's priority);
  * a ctx that was granted the priority gives it up before the lane's next use once another ctx or a pool appears.
This file sorts first among the GPU tests so that its first test still finds a process without any ctx; run after
other GPU tests that test skips itself.  Each test runs its loop ONCE: nothing here tries to provoke the hang.
"""
import ctypes as C

import numpy as np
import pytest

from visual_odometry_amd import _lib, synth
from visual_odometry_amd.device import FrameWorkload

pytestmark = pytest.mark.gpu


def _steps(wl, n):
    for _ in range(n):
        wl.step()
    wl.sync()
    return [t.cpu().numpy().copy() for t in (wl.qa, wl.va, wl.rs_out, wl.rs_inl, wl.tr_X, wl.sel_cnt)]


def _workload(ctx):
    return FrameWorkload(ctx, 300, 300, nv=120, nu=160, n_frames=4, num_hyp=64, levels=3, block=21, keep=30, seed=77)


def test_priority_is_granted_to_a_lone_ctx_and_given_up_when_another_ctx_arrives():
    import torch
    if _lib.Context._default is not None:
        pytest.skip("another vo_ctx already exists in this process: the grant can only be seen in a fresh one")
    a = _lib.Context(0)
    b = None
    try:
        try:
            a.call("vo_ctx_lane_priority", 1)
        except _lib.VoError:
            pytest.skip("this process has held other contexts already (run this file first or alone)")
        wl = _workload(a)
        want = _steps(wl, 6)                       # landmark track + pose chain on the high-priority lane
        wl.drain(); wl.sync()
        b = _lib.Context(0)                        # a second ctx on the GPU: a's lane has to come down
        wl2 = _workload(a)                         # its first lane use replaces the stream (vo_int_lane_ensure)
        got = _steps(wl2, 6)
        for x, y in zip(got, want):
            assert np.array_equal(x, y, equal_nan=True)   # scheduling only: same bits
        wl2.drain(); wl2.sync()
        with pytest.raises(_lib.VoError, match="bad argument|unsupported"):
            a.call("vo_ctx_lane_priority", 1)      # and it cannot be asked for again
        wlb = _workload(b)                         # both contexts work side by side at normal priority
        other = _steps(wlb, 6)
        for x, y in zip(other, want):
            assert np.array_equal(x, y, equal_nan=True)
        wlb.drain(); wlb.sync()
        torch.cuda.synchronize()
    finally:
        if b is not None:
            b.close()
        a.close()


def test_priority_is_refused_after_a_second_ctx_and_the_frame_loop_completes():
    """The order of the log: a second ctx's streams created and destroyed, then the first ctx's frame loop with the
    priority requested."""
    from visual_odometry_amd.frame import FrameLoop
    from oracle import oracle, vo_frame
    first = _lib.Context(0)
    try:
        second = _lib.Context(0)
        wls = _workload(second)                    # its ctx stream, side stream and lane all come to exist
        _steps(wls, 2)
        wls.drain(); wls.sync()
        second.close()
        with pytest.raises(_lib.VoError, match="unsupported") as ei:
            first.call("vo_ctx_lane_priority", 1)
        assert "refused" in str(ei.value)
        K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
        frames, poses, lift = synth.plane_sequence(5, K)
        u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=80)
        st = vo_frame.new_state()
        st.keypoints = np.stack([u, v], 1).astype(np.float64); st.landmarks = lift(u.astype(np.float64), v.astype(np.float64))
        st.reproError = np.zeros(len(u))
        loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=100, ctx=first)
        loop.set_state(st, frames[0])
        statuses = []
        for k in range(1, len(frames)):            # the loop of tests/test_gpu_device_api.py that stood still
            pose, status, info = loop.processFrame(frames[k], frames[k + 1] if k + 1 < len(frames) else None)
            statuses.append(status)
        loop.close()
        assert statuses == [0] * (len(frames) - 1)
    finally:
        first.close()


def test_a_decode_pool_counts_like_a_ctx(tmp_path):
    """vo_ingest_open brings a copy stream: afterwards the priority is refused on that GPU."""
    from visual_odometry_amd import ingest
    c = _lib.Context(0)
    try:
        img = synth.frame_pair(nv=48, nu=64, seed=3)[0]
        p = tmp_path / "a.pgm"
        p.write_bytes(b"P5\n64 48\n255\n" + img.tobytes())
        fs = ingest.FrameStream([str(p)], threads=1, ctx=c)
        assert sum(1 for _ in fs) == 1
        fs.close()
        with pytest.raises(_lib.VoError, match="unsupported"):
            c.call("vo_ctx_lane_priority", 1)
    finally:
        c.close()
