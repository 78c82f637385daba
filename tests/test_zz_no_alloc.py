"""vo_dev_p3p_ransac_static -- the one entry point that allocates nothing (include/vo_hip.h) -- against
vo_dev_p3p_ransac on the same inputs, through the C ABI: identical bytes, on two calls in a row, the second with a
smaller input (the static work array keeps the first call's contents), with and without the adaptive stop; sizes
beyond the array are refused."""
import ctypes as C

import numpy as np
import pytest

from visual_odometry_amd import _lib, synth
from visual_odometry_amd._lib import RansacOpts, ptr

pytestmark = pytest.mark.gpu
K = synth.KITTI_K


def _pose(ctx, torch, entry, xy, X, opts):
    d_xy = torch.from_numpy(np.ascontiguousarray(xy)).cuda(); d_X = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    d_K = torch.from_numpy(np.ascontiguousarray(K)).cuda()
    out = torch.full((16,), -7.0, dtype=torch.float64, device="cuda")
    inl = torch.full((len(xy),), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.call(entry, ptr(d_xy), ptr(d_X), len(xy), ptr(d_K), C.byref(opts), None, None, ptr(out), ptr(inl))
    ctx.sync()
    return out.cpu().numpy().tobytes(), inl.cpu().numpy()


def test_static_entry_point_equals_the_allocating_one(ctx):
    import torch
    big = synth.pnp_scene(3000, seed=51)
    small = synth.pnp_scene(400, seed=52)
    calls = [(big, RansacOpts(4000, 2.0, 0.0, 11, 1, 4)),       # fills most of the static array
             (small, RansacOpts(300, 2.0, 0.0, 12, 1, 4)),      # smaller, right behind it: stale entries beyond 300
             (small, RansacOpts(5000, 2.0, 0.99, 13, 1, 4)),    # the reference's pose step: <= 5000 trials at 99 %
             (big, RansacOpts(64, 1.0, 0.0, 14, 0, 30))]        # plain inlier counting
    got = [_pose(ctx, torch, "vo_dev_p3p_ransac_static", s[0], s[1], o) for s, o in calls]
    want = [_pose(ctx, torch, "vo_dev_p3p_ransac", s[0], s[1], o) for s, o in calls]
    for (gb, gi), (wb, wi) in zip(got, want):
        assert gb == wb and np.array_equal(gi, wi)
    info = np.frombuffer(got[0][0], np.int32, offset=96)
    assert info[0] == 0 and got[0][1].sum() > 2000              # a real pose with its inliers
    with pytest.raises(_lib.VoError, match="VO_STATIC_MAX_HYP|unsupported"):
        _pose(ctx, torch, "vo_dev_p3p_ransac_static", small[0], small[1], RansacOpts(5121, 2.0, 0.0, 1, 1, 4))
    again = _pose(ctx, torch, "vo_dev_p3p_ransac_static", calls[1][0][0], calls[1][0][1], calls[1][1])
    assert again[0] == want[1][0] and np.array_equal(again[1], want[1][1])      # and still right after the refusal


def test_static_entry_point_on_the_staged_path_and_without_any_allocation(ctx):
    """From 4096 points and more than 512 hypotheses on, a run is staged (first hypotheses on every point, the others
    sieved through two survivor lists: k_survivors, the bound and sieve words) -- all of it carved from the static
    array, whose lists keep the previous call's entries.  Both modes, with and without the adaptive stop, twice, with a
    smaller run in between; and the device's free memory does not move across the static calls (inputs and outputs
    are allocated before the first of them)."""
    import torch
    big = synth.pnp_scene(5000, seed=61)
    small = synth.pnp_scene(900, seed=62)
    calls = [(big, RansacOpts(4000, 2.0, 0.0, 21, 1, 4)), (small, RansacOpts(700, 2.0, 0.0, 22, 1, 4)),
             (big, RansacOpts(4000, 2.0, 0.0, 21, 1, 4)), (big, RansacOpts(4000, 2.0, 0.99, 23, 1, 4)),
             (big, RansacOpts(4000, 2.0, 0.0, 24, 0, 30)), (small, RansacOpts(600, 2.0, 0.99, 25, 0, 30)),
             (big, RansacOpts(4000, 2.0, 0.99, 26, 0, 30))]
    want = [_pose(ctx, torch, "vo_dev_p3p_ransac", s[0], s[1], o) for s, o in calls]
    d_K = torch.from_numpy(np.ascontiguousarray(K)).cuda()
    bufs = []
    for (xy, X, *_), o in calls:                   # everything a call touches exists before the first static call
        bufs.append((torch.from_numpy(np.ascontiguousarray(xy)).cuda(), torch.from_numpy(np.ascontiguousarray(X)).cuda(),
                     torch.full((16,), -7.0, dtype=torch.float64, device="cuda"),
                     torch.full((len(xy),), 9, dtype=torch.uint8, device="cuda")))
    torch.cuda.synchronize(); ctx.sync()
    free0 = torch.cuda.mem_get_info()[0]
    for (d_xy, d_X, out, inl), (s, o) in zip(bufs, calls):
        ctx.call("vo_dev_p3p_ransac_static", ptr(d_xy), ptr(d_X), len(s[0]), ptr(d_K), C.byref(o), None, None, ptr(out), ptr(inl))
    ctx.sync()
    assert torch.cuda.mem_get_info()[0] == free0, "a static call changed the device's free memory"
    for (d_xy, d_X, out, inl), (wb, wi) in zip(bufs, want):
        assert out.cpu().numpy().tobytes() == wb and np.array_equal(inl.cpu().numpy(), wi)
    assert want[0][0] == want[2][0]
    info = np.frombuffer(want[0][0], np.int32, offset=96)
    assert info[0] == 0 and want[0][1].sum() > 3000
