"""VisualOdometry_test.m end to end, MATLAB-free: loader -> bootstrap -> device-resident frame loop fed by the
decode pool -> trajectory, evaluated against the loader's ground truth.

    python scripts/run_dataset.py kitti|parking|malaga <path> [num_frames] [bootstrap_frame_0 bootstrap_frame_1]
    python scripts/run_dataset.py synthetic [num_frames]        # writes a parking-layout tree of rendered frames first
"""
import os
import sys
import tempfile

sys.path.insert(0, ".")
import numpy as np

from visual_odometry_amd import ingest, pipeline, synth, trajectory


def write_parking_tree(root, n_frames, nv=240, nu=320):
    """A dataset in the parking layout (utils/dataLoaderParking.m:21,47-50) from a rendered non-planar scene;
    colour PNGs (gray replicated) so that rgb2gray is on the path as with the real parking frames."""
    K = np.array([[300.0, 0, nu / 2], [0, 300.0, nv / 2], [0, 0, 1.0]])
    frames, poses, _ = synth.relief_sequence(n_frames, K, nv=nv, nu=nu, depth=10.0, relief=4.0, step=(0.15, 0.01, 0.05))
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ingest as oi                                # fixture writer only (png_encode)
    for i, f in enumerate(frames):
        with open(os.path.join(root, "images", "img_%05d.png" % i), "wb") as fh:
            fh.write(oi.png_encode(np.repeat(f[:, :, None], 3, 2), [4]))
    with open(os.path.join(root, "K.txt"), "w") as fh:
        for r in K:
            fh.write(", ".join("%.6f" % v for v in r) + ",\n")
    t = trajectory.Trajectory(poses[0])
    for p in poses[1:]:
        t.append(p)
    t.save(os.path.join(root, "poses.txt"))
    return K


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
    if kind == "synthetic":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
        root = tempfile.mkdtemp(prefix="vo_parking_")
        write_parking_tree(root, n + 3)
        loader, num, boot = ingest.dataLoaderParking(root, last_frame=n + 2), n, (0, 2)
    else:
        cls = {"kitti": ingest.dataLoaderKitti, "parking": ingest.dataLoaderParking, "malaga": ingest.dataLoaderMalaga}[kind]
        loader = cls(sys.argv[2])
        num = int(sys.argv[3]) if len(sys.argv) > 3 else 300
        boot = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (20, 21)
    traj, info = pipeline.run_sequence(loader, num_frames=num, bootstrap_frames=boot,
                                       MinNumLandmarks=50 if kind == "synthetic" else 200)
    print(f"bootstrap: {info['bootstrap_landmarks']} landmarks in {1e3 * info['bootstrap_seconds']:.0f} ms (first call: includes warm-up)")
    print(f"{info['frames']} frames in {1e3 * info['seconds']:.1f} ms = {info['frames'] / info['seconds']:.0f} frames/s, "
          f"{sum(s == 0 for s in traj.status)}/{len(traj)} poses ok, landmarks {min(info['n_landmarks'])}..{max(info['n_landmarks'])}")
    gt = loader.ground_truth_data
    if len(gt):
        # pose 0 of the run is the world frame = the first bootstrap view; the others are the frames processed
        g = gt[[boot[0]] + list(range(info["first_frame"], info["first_frame"] + info["frames"]))]
        ev = trajectory.evaluate(traj.topViewTrajectory, g)
        print(f"top view vs ground truth after similarity alignment: rmse {ev['rmse']:.3f} m, max {ev['max']:.3f} m over "
              f"{ev['length']:.1f} m (scale {ev['scale']:.3f})")
    if os.environ.get("VO_PRINT_TRAJECTORY") and len(gt):
        for k in range(len(traj)):
            print(k, traj.status[k], np.round(traj.centres[k], 3), "aligned", np.round(ev["aligned"][k], 3), "truth", np.round(g[k], 3))
    out = os.path.join(tempfile.gettempdir(), "vo_trajectory.txt")
    traj.save(out)
    print("trajectory written to", out)
