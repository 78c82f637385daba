"""Device-resident frame loop over vo_frame_* (include/vo_hip.h): VisualOdometry.processFrame,
modules/VisualOdometry.m:264-346, with the state struct kept in HBM across frames.

    loop = FrameLoop(K, image_shape, maxNumLandmarks=300, ...)      # VisualOdometry.m:22-37 defaults
    loop.set_state(state, prev_img)                                 # e.g. the bootstrap's output
    pose, status, info = loop.processFrame(curr_img)                # :264
    state = loop.get_state()

Coordinates are zero-based (u = column, v = row), K in the same origin; images are row-major uint8
(numpy arrays or device tensors).  There is no CPU fallback.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib
from ._lib import FrameInfo, FrameOpts, KltOpts, RansacOpts, SelOpts, ptr


class FrameLoop:
    def __init__(self, K, image_shape, angularThreshold=1.0, maxTemporalRecall=10, maxNumLandmarks=300,
                 maxReprojectionError=3.0, maxCandidates=2048, NumPyramidLevels=4, MaxBidirectionalError=2.0,
                 BlockSize=(41, 41), MaxIterations=50, MaxNumTrials=5000, Confidence=99.0, PoseTolerance=2.0,
                 MinQuality=0.001, FilterSize=5, MinDistance=7.0, CandidatesToKeep=100, Seed=1023, ctx=None):
        self.ctx = ctx or _lib.Context.default()
        self.nv, self.nu = int(image_shape[0]), int(image_shape[1])
        self.opts = FrameOpts(int(maxNumLandmarks), int(maxCandidates), int(maxTemporalRecall), float(angularThreshold),
                              float(maxReprojectionError),
                              KltOpts(int(NumPyramidLevels), float(MaxBidirectionalError), int(BlockSize[0]), int(BlockSize[1]),
                                      int(MaxIterations), 0.01),
                              RansacOpts(int(MaxNumTrials), float(PoseTolerance), float(Confidence) / 100.0, int(Seed), 1, 4),
                              SelOpts(float(MinDistance), float(MinQuality), int(FilterSize), int(CandidatesToKeep),
                                      _lib.SCORER_MINEIGEN, 0.08))
        self.K = np.ascontiguousarray(K, np.float64)
        self.h = C.c_void_p()
        self.ctx.call("vo_frame_create", C.byref(self.opts), self.nu, self.nv, ptr(self.K), C.byref(self.h))

    def _img(self, img):
        if isinstance(img, np.ndarray):
            a = np.ascontiguousarray(img, np.uint8)
            if a.shape != (self.nv, self.nu):
                raise ValueError("image shape differs from the loop's")
            return a, ptr(a), 0
        if tuple(img.shape) != (self.nv, self.nu) or not img.is_contiguous():
            raise ValueError("device image must be a contiguous nv x nu uint8 tensor")
        return img, C.c_void_p(img.data_ptr()), 1

    def set_image(self, img):
        keep, p, dev = self._img(img)
        self.ctx.call("vo_frame_set_image", self.h, p, dev)

    def set_state(self, st, prev_img=None, frame_no=0):
        L = np.ascontiguousarray(st.landmarks, np.float64).reshape(-1, 3)
        kp = np.ascontiguousarray(st.keypoints, np.float64).reshape(-1, 2)
        rep = np.ascontiguousarray(st.reproError, np.float64).reshape(-1)
        ck = np.ascontiguousarray(st.candidate_keypoints, np.float64).reshape(-1, 2)
        cf = np.ascontiguousarray(st.candidate_first_keypoints, np.float64).reshape(-1, 2)
        cp = np.ascontiguousarray(np.asarray(st.candidate_first_poses, np.float64).reshape(-1, 12))
        ct = np.ascontiguousarray(st.candidate_time_indxs, np.int32).reshape(-1)
        if not (len(L) == len(kp) == len(rep)) or not (len(ck) == len(cf) == len(cp) == len(ct)):
            raise ValueError("state arrays differ in length")
        self.ctx.call("vo_frame_set_state", self.h, ptr(L), ptr(kp), ptr(rep), len(L), ptr(ck), ptr(cf), ptr(cp), ptr(ct),
                      len(ck), int(frame_no))
        if prev_img is not None:
            self.set_image(prev_img)

    def counts(self):
        n, c = C.c_int(), C.c_int()
        self.ctx.call("vo_frame_get_counts", self.h, C.byref(n), C.byref(c))
        return n.value, c.value

    def get_state(self):
        n, c = self.counts()
        L = np.zeros((n, 3)); kp = np.zeros((n, 2)); rep = np.zeros(n)
        ck = np.zeros((c, 2)); cf = np.zeros((c, 2)); cp = np.zeros((c, 4, 3)); ct = np.zeros(c, np.int32)
        self.ctx.call("vo_frame_get_state", self.h, ptr(L), ptr(kp), ptr(rep), ptr(ck), ptr(cf), ptr(cp), ptr(ct))
        return SimpleNamespace(landmarks=L, keypoints=kp, reproError=rep, candidate_keypoints=ck,
                               candidate_first_keypoints=cf, candidate_first_poses=cp,
                               candidate_time_indxs=ct.astype(np.int64))

    def processFrame(self, curr_img, next_img=None):
        """-> (curr_pose 4x3 [R_WC; t_WC], pose_status, info dict); the new state stays on the device.
        next_img: the image the following call will process, if it is at hand already (vo_frame_announce): its
        pyramid and corner detection then run beside this frame's tail.  It must be the very object (same memory)
        that the following call passes, kept unchanged until then.  Same results either way."""
        keep, p, dev = self._img(curr_img)
        if next_img is not None:
            self._next_keep, pn, devn = self._img(next_img)       # kept alive until the next call
            self.ctx.call("vo_frame_announce", self.h, pn, devn)
        pose = np.zeros((4, 3)); info = FrameInfo()
        self.ctx.call("vo_frame_process", self.h, p, dev, ptr(pose), C.byref(info))
        return pose, info.status, {n: getattr(info, n) for n, _ in FrameInfo._fields_}

    MAX_PENDING = 4      # VO_FRAME_MAX_PENDING

    def submit(self, curr_img, next_img=None):
        """Enqueue one processFrame without waiting for it (vo_frame_submit): the host runs ahead of the GPU by up to
        MAX_PENDING frames.  A device image must stay unchanged until its frame has been collected (a reference is
        kept here); next_img as in processFrame."""
        keep, p, dev = self._img(curr_img)
        if next_img is not None:
            self._next_keep, pn, devn = self._img(next_img)
            self.ctx.call("vo_frame_announce", self.h, pn, devn)
        self.ctx.call("vo_frame_submit", self.h, p, dev)
        self._pending = getattr(self, "_pending", [])
        self._pending.append(keep)

    def collect(self):
        """-> (curr_pose, pose_status, info dict) of the OLDEST submitted frame (vo_frame_collect)."""
        pose = np.zeros((4, 3)); info = FrameInfo()
        self.ctx.call("vo_frame_collect", self.h, ptr(pose), C.byref(info))
        if getattr(self, "_pending", None):
            self._pending.pop(0)
        return pose, info.status, {n: getattr(info, n) for n, _ in FrameInfo._fields_}

    def close(self):
        if self.h:
            self.ctx.call("vo_frame_destroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
