"""ctypes loader for libvo_hip.so (the C ABI in include/vo_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is present every
operator raises.  The library is built in-tree by `__graft_entry__.build()` /
`make -C visual_odometry_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(_HERE, "libvo_hip.so")  # VO_HIP_LIB: A/B builds when profiling

VO_OK = 0
SCORER_MINEIGEN, SCORER_HARRIS = 0, 1


class VoError(RuntimeError):
    pass


class Layout(C.Structure):
    _fields_ = [("col_major", C.c_int), ("index_base", C.c_int)]


class KltOpts(C.Structure):
    _fields_ = [("levels", C.c_int), ("max_bidir_err", C.c_double), ("block_h", C.c_int), ("block_w", C.c_int),
                ("max_iters", C.c_int), ("eps", C.c_double)]


class SelOpts(C.Structure):
    _fields_ = [("min_distance", C.c_double), ("min_quality", C.c_double), ("filter_size", C.c_int),
                ("keep", C.c_int), ("scorer", C.c_int), ("kappa", C.c_double)]


class RansacOpts(C.Structure):
    _fields_ = [("num_hyp", C.c_int), ("max_reproj_err", C.c_double), ("confidence", C.c_double),
                ("seed", C.c_uint64), ("mode", C.c_int), ("min_inliers", C.c_int)]


class RansacInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("best_hyp", C.c_int), ("best_sol", C.c_int), ("trials_used", C.c_int),
                ("num_inliers", C.c_int)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("frames", "keypoints_tracked", "keypoints_valid", "ransac_inliers", "elapsed_s")]


class FrameOpts(C.Structure):
    _fields_ = [("max_landmarks", C.c_int), ("max_candidates", C.c_int), ("max_temporal_recall", C.c_int),
                ("angular_threshold_deg", C.c_double), ("max_reproj_err", C.c_double), ("klt", KltOpts),
                ("pose", RansacOpts), ("sel", SelOpts)]


class FrameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("status", "n_landmarks", "n_candidates", "n_tracked", "n_inliers",
                                         "trials_used", "n_promoted", "n_new", "overflow", "frame_no")]


# every symbol include/vo_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "vo_strerror", "vo_version", "vo_ctx_create", "vo_ctx_destroy", "vo_ctx_set_stream", "vo_ctx_get_stream",
    "vo_ctx_sync", "vo_ctx_lane_begin", "vo_ctx_lane_resume", "vo_ctx_lane_end", "vo_ctx_lane_join", "vo_ctx_lane_priority", "vo_dev_frame_wait", "vo_ctx_last_error", "vo_ctx_read_counters", "vo_stats_gather", "vo_timer_start", "vo_timer_stop_ms",
    "vo_timer_lap_begin", "vo_timer_lap_end", "vo_timer_laps_ms", "vo_ctx_time_score", "vo_ctx_read_score_tests",
    "vo_klt_track", "vo_select_candidates", "vo_harris_response", "vo_corner_response",
    "vo_select_keypoints_greedy", "vo_triangulate_candidates", "vo_triangulate_points", "vo_angle_deg", "vo_in_front", "vo_reproj_error",
    "vo_p3p_ransac", "vo_p3p_batch", "vo_essential_ransac", "vo_essential_ransac_five", "vo_dev_frame_build", "vo_dev_klt_track", "vo_dev_klt_track_head", "vo_dev_wait_head", "vo_dev_head_timed_out", "vo_dev_select_candidates", "vo_dev_select_begin", "vo_dev_select_finish",
    "vo_dev_frame_prefetch",
    "vo_dev_triangulate", "vo_dev_p3p_ransac", "vo_dev_p3p_ransac_static", "vo_dev_merge_ranks", "vo_frame_create", "vo_frame_destroy", "vo_frame_set_state",
    "vo_frame_get_counts", "vo_frame_get_state", "vo_frame_set_image", "vo_frame_process", "vo_frame_submit", "vo_frame_collect", "vo_frame_announce",
    "vo_image_info", "vo_image_decode", "vo_rgb2gray", "vo_ingest_open", "vo_ingest_count", "vo_ingest_next",
    "vo_ingest_read", "vo_ingest_stats", "vo_ingest_set_hold", "vo_ingest_close",
]

_lib = None


def load():
    """Load libvo_hip.so (no GPU needed for loading; compute entry points need one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64; if this library pulled in /opt/rocm's copy
        # first, a later `import torch` would bring a second HIP runtime into the process and see no
        # GPU.  Loading torch first makes both bind to one runtime (torch is only plumbing here).
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        lib.vo_strerror.restype = C.c_char_p
        lib.vo_ctx_last_error.restype = C.c_char_p
        lib.vo_ctx_last_error.argtypes = [C.c_void_p]
        lib.vo_ctx_get_stream.restype = C.c_void_p
        lib.vo_ctx_get_stream.argtypes = [C.c_void_p]
        lib.vo_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.vo_ctx_destroy.argtypes = [C.c_void_p]
        lib.vo_ctx_destroy.restype = None
        lib.vo_image_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
        lib.vo_image_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.vo_ingest_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.vo_ingest_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.vo_ingest_close.argtypes = [C.c_void_p]
        lib.vo_ingest_set_hold.argtypes = [C.c_void_p, C.c_int]
        _lib = lib
    return _lib


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return C.c_void_p(a)
    # torch tensor
    return C.c_void_p(a.data_ptr())


class Context:
    """One vo_ctx (one GPU, one stream).  `Context.default()` is a per-process singleton."""
    _default = None

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.vo_ctx_create(int(device), C.byref(h))
        if rc != VO_OK:
            raise VoError(f"vo_ctx_create(device={device}) failed: {self.lib.vo_strerror(rc).decode()}")
        self.h = h
        self.device = device

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = Context(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("VO_USE_LOCAL_RANK") else 0)
        return cls._default

    def check(self, rc, what):
        if rc != VO_OK:
            raise VoError(f"{what} failed: {self.lib.vo_strerror(rc).decode()} "
                          f"[{self.lib.vo_ctx_last_error(self.h).decode()}]")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.h, *args), name)

    def sync(self):
        self.call("vo_ctx_sync")

    def set_stream(self, stream_handle):
        self.call("vo_ctx_set_stream", C.c_void_p(stream_handle))

    def counters(self, reset=False):
        out = (C.c_uint64 * 4)()
        self.call("vo_ctx_read_counters", out, int(reset))
        return list(out)

    def timer_start(self):
        self.call("vo_timer_start")

    def timer_stop_ms(self):
        ms = C.c_float()
        self.call("vo_timer_stop_ms", C.byref(ms))
        return ms.value

    def lap_begin(self):
        self.call("vo_timer_lap_begin")

    def lap_end(self):
        self.call("vo_timer_lap_end")

    def laps_ms(self):
        """-> (sum of the laps in ms, number of laps); one synchronisation, then reset."""
        ms, n = C.c_float(), C.c_int()
        self.call("vo_timer_laps_ms", C.byref(ms), C.byref(n))
        return ms.value, n.value

    def close(self):
        if self.h:
            self.lib.vo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
