"""The N > 1 path on CPU: two gloo ranks run the statistics gather and the sequence assignment."""
import json
import os
import socket
import subprocess
import sys

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
    assert o0["per_rank_elapsed_s"] == [1.0, 2.0]      # ... and every rank's own time is in the record, in rank order
    assert dist.per_rank_ms_per_step(o0, 10) == {"min": 100.0, "max": 200.0, "each": [100.0, 200.0]}
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
    assert out["frames"] == 3 and out["elapsed_s"] == 0.5 and out["world"] == 1 and out["per_rank_elapsed_s"] == [0.5]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns two ranks (before anything touches a
    GPU), they meet over gloo, run the statistics collective, and rank 0's line comes back through the parent."""
    env = dict(os.environ, VO_BENCH_RENDEZVOUS_ONLY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--keypoints", "123"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # one line, from rank 0
    out = json.loads(lines[0])
    assert out["rendezvous"] == "ok" and out["n_gpus"] == 2 and out["frames"] == 2 and out["elapsed_s"] == 2.0
    assert out["per_rank_ms_per_step"] == {"min": 1000.0, "max": 2000.0, "each": [1000.0, 2000.0]}   # the line shows stragglers


def test_a_failing_rank_fails_the_launch(tmp_path):
    """One rank exits non-zero before the rendezvous: the launcher stops the other one and reports failure."""
    from visual_odometry_amd import launch
    script = tmp_path / "ranks.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "time.sleep(60)\n")
    import time
    t0 = time.time()
    assert launch.spawn_ranks(str(script), [], 2, grace_s=2.0) == 3
    assert time.time() - t0 < 30
