"""Independent sequences over the GPUs of a node, one process per GPU (BASELINE configs[4]: KITTI 00-10 over
8 GPUs): longest-first assignment by frame count (dist.assign_sequences), every rank runs its sequences end to
end (loader -> bootstrap -> decode pool -> device-resident frame loop -> trajectory), one collective at the end
for the statistics.  No data-path exchange: frames of a sequence are a Markov chain.

    python scripts/run_sharded.py --gpus N kitti <root with 00..10/image_0 and poses/> [frames per sequence]
    python scripts/run_sharded.py --gpus N synthetic <n_sequences> [frames per sequence]   # rendered parking-layout trees
(the script starts its N ranks itself, visual_odometry_amd/launch.py; under torch.distributed.run it takes the
launcher's ranks instead and --gpus is not needed)
VO_REHEARSE_ONE_GPU=1 puts every rank on device 0 with gloo (a rehearsal of the N > 1 path on a one-GPU box).
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visual_odometry_amd import launch

if "--gpus" in sys.argv:                            # start the ranks from here, before torch or the GPU is touched
    k = sys.argv.index("--gpus")
    n_gpus = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
    if not launch.under_launcher():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], n_gpus))

rehearse = bool(os.environ.get("VO_REHEARSE_ONE_GPU"))
if not rehearse:
    os.environ["VO_USE_LOCAL_RANK"] = "1"          # Context.default() = this rank's GPU
else:
    os.environ.setdefault("VO_LANE_NORMAL_PRIORITY", "1")   # several processes on one GPU: see bench.py

import numpy as np

from visual_odometry_amd import dist, ingest, pipeline, trajectory

rank, local_rank, world = dist.init("gloo" if rehearse else None)
kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
if kind == "synthetic":
    n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import run_dataset
    lengths = [frames + 4 * (i % 3) for i in range(n_seq)]            # unequal, so that the assignment matters
    mine = dist.assign_sequences(lengths, world)[rank]
    loaders = []
    for i in mine:
        root = tempfile.mkdtemp(prefix="vo_seq%02d_" % i)
        run_dataset.write_parking_tree(root, lengths[i] + 3)
        loaders.append((i, ingest.dataLoaderParking(root, last_frame=lengths[i] + 2), lengths[i], (0, 2), 50))
else:
    root = sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
    seqs = sorted(d for d in os.listdir(root) if d.isdigit() and os.path.isdir(os.path.join(root, d, "image_0")))
    counts = [len(os.listdir(os.path.join(root, s, "image_0"))) for s in seqs]
    lengths = [min(c - 22, frames) for c in counts]
    mine = dist.assign_sequences(lengths, world)[rank]
    loaders = [(i, ingest.dataLoaderKitti(root, last_frame=counts[i] - 1, sequence=seqs[i]), lengths[i], (20, 21), 200) for i in mine]

t0 = time.perf_counter()
done, ok_poses, report = 0, 0, []
for i, loader, n, boot, min_lm in loaders:
    traj, info = pipeline.run_sequence(loader, num_frames=n, bootstrap_frames=boot, MinNumLandmarks=min_lm)
    done += info["frames"]; ok_poses += sum(s == 0 for s in traj.status[1:])
    gt = loader.ground_truth_data
    ev = trajectory.evaluate(traj.topViewTrajectory, gt[[boot[0]] + list(range(info["first_frame"], info["first_frame"] + info["frames"]))])
    report.append({"sequence": i, "frames": info["frames"], "frames_per_s": info["frames"] / info["seconds"],
                   "rmse_m": ev["rmse"], "path_m": ev["length"]})
elapsed = time.perf_counter() - t0
dist.barrier()
stats = dist.gather_stats({"frames": done, "keypoints_tracked": 0, "keypoints_valid": 0, "ransac_inliers": ok_poses, "elapsed_s": elapsed})
for r in range(world):
    if r == rank:
        for line in report:
            print("rank %d" % rank, json.dumps(line), flush=True)
    dist.barrier()
if rank == 0:
    print(json.dumps({"sequences": len(lengths), "n_gpus": world, "frames": stats["frames"], "poses_ok": stats["ransac_inliers"],
                      "elapsed_s": stats["elapsed_s"], "frames_per_s": stats["frames"] / stats["elapsed_s"],
                      "note": "elapsed = slowest rank, bootstrap and decode included"}))
dist.shutdown()
