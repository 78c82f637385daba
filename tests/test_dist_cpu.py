"""The N > 1 path on CPU: two gloo ranks run the statistics gather and the sequence assignment."""
import os
import socket

import torch.multiprocessing as mp

from visual_odometry_amd import dist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = dist.init(backend="gloo")
    mine = dist.assign_sequences([4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201], w)[r]
    dist.barrier()
    out = dist.gather_stats({"frames": 10 * (r + 1), "keypoints_tracked": 1000 * (r + 1), "keypoints_valid": 900,
                             "ransac_inliers": 5, "elapsed_s": 1.0 + r})
    q.put((r, mine, out))
    dist.shutdown()


def test_two_rank_gloo_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, s0, o0), (r1, s1, o1) = res
    assert o0 == o1                                    # every rank sees the same gathered record
    assert o0["frames"] == 30 and o0["keypoints_tracked"] == 3000 and o0["keypoints_valid"] == 1800
    assert o0["elapsed_s"] == 2.0 and o0["world"] == 2  # the job ends with its slowest rank
    assert sorted(s0 + s1) == list(range(11)) and not set(s0) & set(s1)


def test_assignment_is_balanced_and_deterministic():
    kitti = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]   # frames of KITTI 00-10
    a = dist.assign_sequences(kitti, 8)
    assert a == dist.assign_sequences(kitti, 8)
    loads = [sum(kitti[i] for i in r) for r in a]
    assert max(loads) == 4661 and all(len(r) >= 1 for r in a)                   # bound by the longest sequence
    assert dist.assign_sequences(kitti, 1) == [sorted(range(11), key=lambda i: (-kitti[i], i))]


def test_single_process_gather_is_identity():
    out = dist.gather_stats({"frames": 3, "elapsed_s": 0.5})
    assert out["frames"] == 3 and out["elapsed_s"] == 0.5 and out["world"] == 1
