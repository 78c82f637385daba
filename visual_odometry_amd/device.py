"""Device-resident frame workload over the vo_dev_* entry points (inputs and outputs stay in HBM).

One `step` is the hot path of one VO frame (modules/VisualOdometry.m:264-346) on synthetic inputs:
pyramid build of the new frame (the previous frame's pyramid is reused from its slot), KLT track of
the landmark keypoints (:278) and of the candidate keypoints (:71), P3P-RANSAC pose (:285),
batched candidate triangulation + angle + cheirality (:87-113) and candidate selection (:330).
torch is used for device memory only.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, synth
from ._lib import RansacOpts, SelOpts


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class FrameWorkload:
    def __init__(self, ctx, n_landmarks, n_candidates, nv=376, nu=1241, n_frames=16, num_hyp=2000, levels=4,
                 block=41, max_iters=50, eps=0.01, max_bidir=2.0, keep=100, seed=1023, device=None, ahead=None,
                 lane_priority=False, head=None):
        """ahead: the image-only work of a frame (pyramid, corner detection) is issued one step ahead on the side
        stream (vo_dev_frame_prefetch); None: on unless VO_NO_PREFETCH is set (A/B switch for profiling)."""
        self.ctx = ctx
        self.ahead = (not os.environ.get("VO_NO_PREFETCH")) if ahead is None else bool(ahead)
        # branch-parallel schedule: landmark tracking -> pose on the ctx's lane stream beside candidate tracking
        # (modules/VisualOdometry.m: the pose step :285-289 needs the landmark tracks of :278 only, the candidate
        # tracking of :71 neither).  VO_NO_LANE=1: one tracking launch over both sets, then the pose (A/B switch).
        self.lane = not os.environ.get("VO_NO_LANE")
        self.scores = bool(os.environ.get("VO_KLT_SCORES"))      # the frame step discards the tracker's scores
        # head schedule (default): both point sets in ONE tracking launch with the landmarks as its head -- dispatched
        # first, the pose chain on the lane released by the last of them (vo_dev_klt_track_head / vo_dev_wait_head): the
        # order of the blocks does what the stream priority did, at any priority.  VO_NO_HEAD=1: two launches on two
        # streams (round 2's form, which needs the high-priority lane to start the pose chain early).
        self.head = (not os.environ.get("VO_NO_HEAD")) if head is None else bool(head)
        # lane_priority: the lane as a high-priority stream (vo_ctx_lane_priority) -- for a process with one ctx on a GPU of
        # its own (bench.py); refused, and ignored here, once the ctx's lane exists
        if lane_priority and self.lane:
            try:
                ctx.call("vo_ctx_lane_priority", 1)
            except Exception:
                pass
        self.dev = torch.device("cuda", ctx.device if device is None else device)
        self.nu, self.nv, self.levels, self.block = nu, nv, levels, block
        self.max_iters, self.eps, self.max_bidir = max_iters, eps, max_bidir
        self.nl, self.nc = int(n_landmarks), int(n_candidates)
        frames, self.warps = synth.frame_sequence(n_frames + 1, nv, nu, seed=seed)
        self.frames = [torch.from_numpy(f).to(self.dev) for f in frames]          # uint8, resident
        self.n_frames = n_frames
        f64 = dict(dtype=torch.float64, device=self.dev)

        def pts(n, sd):
            u, v = synth.keypoints(n, nv, nu, seed=sd)
            return torch.from_numpy(np.stack([u, v])).to(self.dev).contiguous()       # SoA 2 x n

        # landmark and candidate keypoints live back to back so that the frame's two track calls
        # (modules/VisualOdometry.m:278 and :71) can be issued as ONE launch over both sets
        self.pa = torch.cat([pts(self.nl, seed + 1024), pts(self.nc, seed + 2048)], 1).contiguous()
        self.qa = torch.empty_like(self.pa)
        self.va = torch.empty(self.nl + self.nc, dtype=torch.uint8, device=self.dev)
        self.sa = torch.zeros(self.nl + self.nc, dtype=torch.float32, device=self.dev)   # stays zero when the scores are not asked for
        self.pl, self.pc = self.pa[:, :self.nl], self.pa[:, self.nl:]
        self.ql, self.qc = self.qa[:, :self.nl], self.qa[:, self.nl:]
        self.vl, self.vc = self.va[:self.nl], self.va[self.nl:]
        self.sl, self.sc = self.sa[:self.nl], self.sa[self.nl:]
        # pose step: landmarks + observations
        xy, X, *_ = synth.pnp_scene(self.nl, seed=seed + 3072, nv=nv, nu=nu)
        self.rs_xy = torch.from_numpy(xy).to(self.dev); self.rs_X = torch.from_numpy(X).to(self.dev)
        self.K = torch.from_numpy(np.ascontiguousarray(synth.KITTI_K)).to(self.dev)
        self.rs_out = torch.zeros(16, **f64)
        self.rs_inl = torch.empty(self.nl, dtype=torch.uint8, device=self.dev)
        self.rs_opts = RansacOpts(int(num_hyp), 2.0, 0.0, seed, 1, 4)
        # triangulation: candidates observed first from one of 10 poses (pose table + index)
        xy0, xy1, poses0, pose1, _ = synth.triangulation_scene(self.nc, seed=seed + 4096, nv=nv, nu=nu)
        uniq, idx = np.unique(poses0.reshape(len(poses0), -1), axis=0, return_inverse=True)
        self.tr_xy0 = torch.from_numpy(xy0).to(self.dev); self.tr_xy1 = torch.from_numpy(xy1).to(self.dev)
        self.tr_poses = torch.from_numpy(np.ascontiguousarray(uniq)).to(self.dev)
        self.tr_idx = torch.from_numpy(idx.astype(np.int32)).to(self.dev)
        self.tr_pose1 = torch.from_numpy(np.ascontiguousarray(pose1)).to(self.dev)
        self.tr_X = torch.empty((self.nc, 3), **f64)
        self.tr_rep = torch.empty(self.nc, **f64); self.tr_ang = torch.empty(self.nc, **f64)
        self.tr_val = torch.empty(self.nc, dtype=torch.uint8, device=self.dev)
        # selection against every tracked point
        ex = self.pa.to(torch.float32).contiguous()
        self.ex = ex; self.n_exist = ex.shape[1]
        self.sel_opts = SelOpts(7.0, 0.001, 5, int(keep), _lib.SCORER_MINEIGEN, 0.08)
        self.sel_u = torch.empty(max(keep, 1), dtype=torch.float32, device=self.dev)
        self.sel_v = torch.empty_like(self.sel_u)
        self.sel_lin = torch.empty(max(keep, 1), dtype=torch.int32, device=self.dev)
        self.sel_cnt = torch.zeros(1, dtype=torch.int32, device=self.dev)
        torch.cuda.synchronize(self.dev)
        self.k = 0
        if self.ahead:
            self.build(3, 0)           # frame 0 as "previous" of step 0
            self.prefetch(0, 1)        # frame 1: what step 0 tracks into and selects on
        else:
            self.build(0, 0)

    # ---- single stages (asynchronous on the ctx stream) ----
    def build(self, slot, frame):
        self.ctx.call("vo_dev_frame_build", slot, _p(self.frames[frame]), self.nu, self.nv, self.levels)

    def track(self, a, b, p, q, valid, score):
        """score: None = not wanted, as in the frame step the workload restates -- modules/VisualOdometry.m:278 and :71
        take `[points, validity, ~]` from the tracker (VO_KLT_SCORES=1 computes them all the same: A/B switch)"""
        n = p.shape[1]
        self.ctx.call("vo_dev_klt_track", a, b, _p(p[0]), _p(p[1]), n, self.block, self.block, self.max_iters,
                      C.c_double(self.eps), C.c_double(self.max_bidir), _p(q[0]), _p(q[1]), _p(valid), _p(score), None)

    def track_all(self, a, b):
        self.track(a, b, self.pa, self.qa, self.va, self.sa if self.scores else None)

    def track_head(self, a, b):
        """landmarks + candidates in one launch, the landmarks as its head"""
        self.ctx.call("vo_dev_klt_track_head", a, b, _p(self.pa[0]), _p(self.pa[1]), self.nl, self.nl + self.nc, self.block,
                      self.block, self.max_iters, C.c_double(self.eps), C.c_double(self.max_bidir), _p(self.qa[0]), _p(self.qa[1]),
                      _p(self.va), _p(self.sa if self.scores else None), None)

    def track_landmarks(self, a, b):
        self.track(a, b, self.pl, self.ql, self.vl, self.sl if self.scores else None)

    def track_candidates(self, a, b):
        self.track(a, b, self.pc, self.qc, self.vc, self.sc if self.scores else None)

    def pose(self):
        self.ctx.call("vo_dev_p3p_ransac", _p(self.rs_xy), _p(self.rs_X), self.nl, _p(self.K), C.byref(self.rs_opts),
                      None, None, _p(self.rs_out), _p(self.rs_inl))

    def triangulate(self):
        self.ctx.call("vo_dev_triangulate", _p(self.tr_xy0), _p(self.tr_xy1), _p(self.tr_poses),
                      self.tr_poses.shape[0], _p(self.tr_idx), self.nc, _p(self.tr_pose1), _p(self.K), _p(self.tr_X),
                      _p(self.tr_rep), _p(self.tr_ang), _p(self.tr_val))

    def select(self, frame):
        self.ctx.call("vo_dev_select_candidates", _p(self.frames[frame]), self.nu, self.nv, 0, _p(self.ex[0]),
                      _p(self.ex[1]), self.n_exist, C.byref(self.sel_opts), _p(self.sel_u), _p(self.sel_v),
                      _p(self.sel_lin), None, _p(self.sel_cnt))

    def select_begin(self, frame):
        """The image-only half of the selection, on the ctx's side stream beside what is enqueued next."""
        self.ctx.call("vo_dev_select_begin", _p(self.frames[frame]), self.nu, self.nv, C.byref(self.sel_opts), self.n_exist)

    def select_finish(self):
        self.ctx.call("vo_dev_select_finish", 0, _p(self.ex[0]), _p(self.ex[1]), self.n_exist, _p(self.sel_u), _p(self.sel_v),
                      _p(self.sel_lin), None, _p(self.sel_cnt))

    def prefetch(self, slot, frame, select=True):
        """pyramid of `frame` into `slot` (+ the detect half of the selection on it) on the side stream."""
        self.ctx.call("vo_dev_frame_prefetch", slot, _p(self.frames[frame]), self.nu, self.nv, self.levels,
                      C.byref(self.sel_opts) if select else None, self.n_exist)

    def step_ahead(self, lap=False):
        """One step with the next step's image-only work issued beside this step's tail.  Step k tracks frame
        k % n -> k % n + 1; the new frame's pyramid sits in slot k % 3 (prefetched during step k-1), the previous
        frame's in slot (k-1) % 3, or in slot 3 when the sequence has just wrapped round to frame 0."""
        k, n = self.k, self.n_frames
        a = 3 if k % n == 0 else (k - 1) % 3
        b = k % 3
        branch = self.lane and not lap and self.nl >= 4
        if branch and self.head:
            self.ctx.call("vo_dev_frame_wait", a); self.ctx.call("vo_dev_frame_wait", b)
            self.ctx.call("vo_ctx_lane_begin")   # the lane forks HERE, in front of the tracking launch ...
            self.ctx.call("vo_ctx_lane_end")
            self.track_head(a, b)                # ... which is enqueued before the kernel that waits for its head
            self.ctx.call("vo_ctx_lane_resume")
            self.ctx.call("vo_dev_wait_head")
            self.pose()
            self.ctx.call("vo_ctx_lane_end")
        elif branch:
            self.ctx.call("vo_dev_frame_wait", a); self.ctx.call("vo_dev_frame_wait", b)
            self.ctx.call("vo_ctx_lane_begin")
            self.track_landmarks(a, b)
            self.ctx.call("vo_ctx_lane_end")
            self.track_candidates(a, b)          # enqueued before the pose chain's dozen launches: both streams fed early
            self.ctx.call("vo_ctx_lane_resume")
            self.pose()
            self.ctx.call("vo_ctx_lane_end")
        else:
            if lap:
                self.ctx.lap_begin()
            self.track_all(a, b)
            if lap:
                self.ctx.lap_end()
        k1 = k + 1                       # the step after this one: its new frame, and frame 0 again if it wraps
        self.prefetch(k1 % 3, (k1 % n) + 1)
        if k1 % n == 0:
            self.prefetch(3, 0, select=False)
        if branch:
            self.ctx.call("vo_ctx_lane_join")
        elif self.nl >= 4:
            self.pose()
        self.triangulate()
        self.select_finish()             # the selection on this step's new frame, begun one step ago
        self.k += 1

    def drain(self):
        """Finish what was begun ahead (the open selection), so that single stages can be called on their own."""
        if self.ahead and self.k >= 0:
            self.select_finish()
            self.ahead = False
            self.build(self.k % 2, self.k % self.n_frames)     # the serial steps' "previous" frame

    def step(self, lap=False):
        """frame k (slot k%2, already built) -> frame k+1.  lap: a vo_timer lap around the tracking launch."""
        if self.ahead:
            return self.step_ahead(lap)
        k = self.k
        a, b = k % 2, (k + 1) % 2
        nxt = (k % self.n_frames) + 1
        # the selection in two calls (its image-only half beside tracking) unless switched off for an A/B, or a lap
        # is wanted: the lap would also count the time k_lk queues behind that side-stream work
        two_phase = not os.environ.get("VO_SINGLE_CALL_SELECT") and not lap
        if two_phase:
            self.select_begin(nxt)     # corner detection on the new image runs beside the pyramid build and tracking
        self.build(b, nxt)
        if lap:
            self.ctx.lap_begin()
        self.track_all(a, b)       # landmarks + candidates in one launch
        if lap:
            self.ctx.lap_end()
        if self.nl >= 4:
            self.pose()
        self.triangulate()
        if two_phase:
            self.select_finish()       # distance filter against the existing points + selectUniform
        else:
            self.select(nxt)
        self.k += 1
        if self.k % self.n_frames == 0:       # loop the sequence: rebuild frame 0 as "previous"
            self.build(self.k % 2, 0)

    def sync(self):
        self.ctx.sync()
