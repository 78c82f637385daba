"""A whole run over a dataset, as module_tests/VisualOdometry_test.m:25-103 does it: bootstrap on two frames of
the loader (:29-43), then `processFrame` over the following frames (:88-103) with the poses collected into a
Trajectory (:72-74,101).  Everything per frame runs on the device: the loader's FrameStream decodes ahead and
uploads, FrameLoop keeps the state in HBM (SURVEY 8f ranks 1-4 end to end).

MATLAB's pixel coordinates are 1-based; the frame loop's are zero-based with K in the same origin, so the
bootstrap runs with the principal point shifted by one (as scripts/run_vo.py does) and its keypoints are
shifted back.
"""
import time

import numpy as np

from . import ingest, modules
from .frame import FrameLoop
from .trajectory import Trajectory


def _gray(img):
    return ingest.rgb2gray(img) if img.ndim == 3 else img


def run_sequence(loader, num_frames=300, bootstrap_frames=(20, 21), MaxNumLandmarks=500, MaxReprojectionError=2.0,
                 MaxTemporalRecall=10, MinNumLandmarks=200, MaxDepth=200.0, MinQuality=0.005, threads=0, depth=0,
                 **loop_options):
    """-> (Trajectory, info).  Defaults are the script's (VisualOdometry_test.m:29,37-41,56,61-66)."""
    K = np.asarray(loader.camParams.K, np.float64)
    img0 = _gray(loader.retrieveFrame(int(bootstrap_frames[0]))[0])
    img1 = _gray(loader.retrieveFrame(int(bootstrap_frames[1]))[0])
    K1 = K.copy(); K1[0, 2] += 1.0; K1[1, 2] += 1.0
    t0 = time.perf_counter()
    kp, X, rep, pose = modules.bootstrap(img0, img1, modules.CameraParams(K1), MinNumLandmarks=MinNumLandmarks,
                                         MaxDepth=MaxDepth, FeatureMatchingMode="KLT", FilterSize=3,
                                         MinQuality=MinQuality)                                  # :37-41
    t_boot = time.perf_counter() - t0
    best = np.argsort(rep, kind="stable")[:int(MaxNumLandmarks)]
    st = modules._new_state()
    st.keypoints, st.landmarks, st.reproError = kp[best] - 1.0, X[best], rep[best]
    st.candidate_first_poses = np.zeros((0, 4, 3))
    loop = FrameLoop(K, img1.shape, maxNumLandmarks=MaxNumLandmarks, maxReprojectionError=MaxReprojectionError,
                     maxTemporalRecall=MaxTemporalRecall, **loop_options)                        # :63-66
    loop.set_state(st, img1)
    first = int(bootstrap_frames[1]) + 1
    loader.reset(first)                                                                          # :43
    last = min(first + int(num_frames), loader.last_frame + 1)
    traj = Trajectory()                                                                          # :72-74
    fs = loader.stream(first, last, threads=threads, depth=depth, hold=2)
    n_landmarks = []
    t0 = time.perf_counter()
    try:
        img = next(fs, None)
        while img is not None:
            nxt = next(fs, None)        # the pool runs ahead: the frame after this one is announced to the loop
            p, status, info = loop.processFrame(img, nxt)                                        # :94-95
            traj.append(p, status)                                                               # :101
            n_landmarks.append(info["n_landmarks"])
            img = nxt
    finally:
        dt = time.perf_counter() - t0
        fs.close()
        loop.close()
    loader.reset(last if last <= loader.last_frame else loader.last_frame)
    return traj, {"bootstrap_landmarks": len(kp), "bootstrap_seconds": t_boot, "bootstrap_pose": pose,
                  "frames": len(traj) - 1, "seconds": dt, "first_frame": first, "n_landmarks": n_landmarks}
