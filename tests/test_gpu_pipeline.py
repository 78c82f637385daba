"""A whole run as module_tests/VisualOdometry_test.m:25-103 does it, MATLAB-free (SURVEY 8f ranks 1-4 together):
loader (parking layout, colour PNG) -> bootstrap -> decode pool -> device-resident frame loop -> trajectory,
against the rendered scene's ground truth and against the same run fed from host arrays."""
import importlib.util
import os

import numpy as np
import pytest

from visual_odometry_amd import ingest, modules, pipeline, trajectory
from visual_odometry_amd.frame import FrameLoop

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dataset_run_end_to_end(ctx, tmp_path):
    spec = importlib.util.spec_from_file_location("run_dataset", os.path.join(ROOT, "scripts", "run_dataset.py"))
    rd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rd)
    n = 14
    rd.write_parking_tree(str(tmp_path), n + 3)
    loader = ingest.dataLoaderParking(tmp_path, last_frame=n + 2)
    traj, info = pipeline.run_sequence(loader, num_frames=n, bootstrap_frames=(0, 2), MinNumLandmarks=50, threads=3)
    assert info["frames"] == n and len(traj) == n + 1 and traj.poses.shape == (4, 3, n + 1)
    assert all(s == 0 for s in traj.status) and min(info["n_landmarks"]) >= 100
    gt = loader.ground_truth_data[[0] + list(range(3, 3 + n))]
    ev = trajectory.evaluate(traj.topViewTrajectory, gt)
    assert ev["rmse"] < 0.1 * ev["length"], ev                      # monocular, minimal-solver poses: a sanity bound
    R = traj.poses[:3, :, -1]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-9)

    # the same frames uploaded from host arrays give the same poses, bit for bit
    img = [ingest.rgb2gray(ingest.imread(loader._path(i))) for i in range(n + 3)]
    K = loader.camParams.K
    K1 = K.copy(); K1[0, 2] += 1; K1[1, 2] += 1
    kp, X, rep, _ = modules.bootstrap(img[0], img[2], modules.CameraParams(K1), MinNumLandmarks=50, MaxDepth=200.0,
                                      FilterSize=3, MinQuality=0.005)
    best = np.argsort(rep, kind="stable")[:500]
    st = modules._new_state()
    st.keypoints, st.landmarks, st.reproError = kp[best] - 1.0, X[best], rep[best]
    st.candidate_first_poses = np.zeros((0, 4, 3))
    loop = FrameLoop(K, img[0].shape, maxNumLandmarks=500, maxReprojectionError=2.0, maxTemporalRecall=10)
    loop.set_state(st, img[2])
    for k in range(3, n + 3):
        p, status, _ = loop.processFrame(img[k])
        assert np.array_equal(p, traj.poses[:, :, k - 2]), k
    loop.close()


def test_malaga_layout_run_through_the_jpeg_path(ctx, tmp_path):
    """BASELINE configs[2]'s shape of problem on a rendered scene: a Malaga-layout tree (left / right JPEG files
    interleaved, utils/dataLoaderMalaga.m:54-57), Huffman decoding on the pool's threads, IDCT / upsampling /
    colour / rgb2gray on the device, then bootstrap + the frame loop with its seeded hypothesis stream: the run
    is reproducible bit for bit and equals the same run fed from host-decoded images."""
    PIL_Image = pytest.importorskip("PIL.Image")
    import io
    from visual_odometry_amd import synth
    n = 10
    K = ingest.MALAGA_K * np.array([[0.4, 1, 0.4], [1, 0.4, 0.4], [1, 1, 1]])          # the camera scaled to 320 x 240
    frames, poses, _ = synth.relief_sequence(n + 3, K, nv=240, nu=320, depth=10.0, relief=4.0, step=(0.15, 0.01, 0.05))
    d = tmp_path / ingest.MALAGA_DIR
    d.mkdir()
    for i, f in enumerate(frames):
        for side in ("left", "right"):
            img = np.repeat((f if side == "left" else f[:, ::-1])[:, :, None], 3, 2)
            b = io.BytesIO()
            PIL_Image.fromarray(img).save(b, "JPEG", quality=95, subsampling=2)
            (d / ("img_CAMERA1_%010.3f_%s.jpg" % (1261230000 + 0.05 * i, side))).write_bytes(b.getvalue())

    class Loader(ingest.dataLoaderMalaga):                                           # the extract's K is for 800 x 600
        def __init__(self, path):
            super().__init__(path)
            self.camParams = modules.CameraParams(K)
    runs = []
    for _ in range(2):
        loader = Loader(tmp_path)
        assert loader.last_frame == n + 2 and all("_left" in p for p in loader.left_images_dirs)
        traj, info = pipeline.run_sequence(loader, num_frames=n, bootstrap_frames=(0, 2), MinNumLandmarks=50, threads=2)
        assert info["frames"] == n and all(s == 0 for s in traj.status) and min(info["n_landmarks"]) >= 100
        runs.append(traj.poses)
    assert np.array_equal(runs[0], runs[1])                                          # seeded: reproducible
    # host-decoded images through the same loop
    loader = Loader(tmp_path)
    img = [ingest.rgb2gray(ingest.imread(loader._path(i))) for i in range(n + 3)]
    K1 = K.copy(); K1[0, 2] += 1; K1[1, 2] += 1
    kp, X, rep, _ = modules.bootstrap(img[0], img[2], modules.CameraParams(K1), MinNumLandmarks=50, MaxDepth=200.0,
                                      FilterSize=3, MinQuality=0.005)
    best = np.argsort(rep, kind="stable")[:500]
    st = modules._new_state()
    st.keypoints, st.landmarks, st.reproError = kp[best] - 1.0, X[best], rep[best]
    st.candidate_first_poses = np.zeros((0, 4, 3))
    loop = FrameLoop(K, img[0].shape, maxNumLandmarks=500, maxReprojectionError=2.0, maxTemporalRecall=10)
    loop.set_state(st, img[2])
    for k in range(3, n + 3):
        p, status, _ = loop.processFrame(img[k])
        assert np.array_equal(p, runs[0][:, :, k - 2]), k
    loop.close()
