"""Multi-GPU harness helpers: one process per GPU, independent sequences per rank, and ONE small
collective that gathers run statistics (RCCL on GPUs -- torch's "nccl" backend -- or gloo on CPU).

Frames of one sequence are a Markov chain (modules/VisualOdometry.m:264-281: frame k needs the state
of frame k-1), so a sequence never spans GPUs; the path shards across sequences / synthetic frame
batches with no data-path exchange.
"""
import os

import torch
import torch.distributed as dist

STAT_KEYS = ("frames", "keypoints_tracked", "keypoints_valid", "ransac_inliers", "elapsed_s")


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when run directly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def assign_sequences(lengths, world):
    """Longest-processing-time-first assignment of sequences (by frame count) to ranks, e.g. KITTI
    00-10 over 8 GPUs (BASELINE.json configs[4]).  Returns a list of index lists, one per rank;
    deterministic (ties: lower sequence index first, lower rank first)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


def _device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    if dist.is_initialized():
        dist.barrier()


def gather_stats(local):
    """Sum the counters over ranks and take the MAX of elapsed_s (the job ends with its slowest
    rank).  `local` maps STAT_KEYS to numbers; every rank gets the same dict back.  Beside the maximum the record
    carries every rank's own elapsed_s in rank order ("per_rank_elapsed_s": one all_gather of a double inside this
    same collective), so that a scaling run shows its stragglers."""
    vals = torch.tensor([float(local.get(k, 0.0)) for k in STAT_KEYS], dtype=torch.float64, device=_device())
    each = [float(local.get("elapsed_s", 0.0))]
    if dist.is_initialized():
        mine = vals[STAT_KEYS.index("elapsed_s")].reshape(1).clone()
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        each = [float(p.item()) for p in parts]
        sums = vals.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        vals = sums
        vals[STAT_KEYS.index("elapsed_s")] = mx[STAT_KEYS.index("elapsed_s")]
    out = {k: float(v) for k, v in zip(STAT_KEYS, vals.tolist())}
    out["world"] = dist.get_world_size() if dist.is_initialized() else 1
    out["per_rank_elapsed_s"] = each
    return out


def per_rank_ms_per_step(stats, steps):
    """what bench.py's line carries for N > 1: every rank's own time per step, and their extremes"""
    each = [1e3 * e / steps for e in stats["per_rank_elapsed_s"]]
    return {"min": min(each), "max": max(each), "each": each}


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
