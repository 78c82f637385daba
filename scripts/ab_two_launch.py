"""One 50k-keypoint tracking launch against two 25k launches on two streams (lane), nothing else on the chip."""
import sys, time
sys.path.insert(0, ".")
import torch
from visual_odometry_amd import _lib
from visual_odometry_amd.device import FrameWorkload
ctx = _lib.Context(0)
wl = FrameWorkload(ctx, 25000, 25000, n_frames=2, ahead=False)
wl.build(1, 1)
def one():
    wl.track_all(0, 1)
def two():
    ctx.call("vo_ctx_lane_begin"); wl.track_landmarks(0, 1); ctx.call("vo_ctx_lane_end")
    wl.track_candidates(0, 1)
    ctx.call("vo_ctx_lane_join")
def two_pose():
    ctx.call("vo_ctx_lane_begin"); wl.track_landmarks(0, 1); ctx.call("vo_ctx_lane_end")
    wl.track_candidates(0, 1)
    ctx.call("vo_ctx_lane_resume"); wl.pose(); ctx.call("vo_ctx_lane_end")
    ctx.call("vo_ctx_lane_join")
def one_pose():
    wl.track_all(0, 1); wl.pose()
for name, f in (("one", one), ("two", two), ("two+pose", two_pose), ("one+pose", one_pose), ("one", one), ("two", two)):
    for _ in range(5): f()
    ctx.sync(); t = time.perf_counter()
    for _ in range(40): f()
    ctx.sync(); print(name, "%.4f ms" % ((time.perf_counter() - t) / 40 * 1e3))
