"""Lifetimes: contexts, frame loops and decode pools opened and closed many times give back what they took
(device memory by hipMemGetInfo through torch, host memory by the resident set), and a context of its own
computes the same bits as the session's."""
import importlib.util
import os

import numpy as np
import pytest

from visual_odometry_amd import _lib, ingest, ops, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_used():
    import torch
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    return total - free


def host_rss():
    import psutil
    return psutil.Process().memory_info().rss


def test_contexts_give_back_their_memory(ctx):
    img1, img2, _ = synth.frame_pair(nv=240, nu=320, seed=51, shift=(1.1, 0.6))
    pu, pv = synth.keypoints(500, nv=240, nu=320, seed=52)
    pts = np.stack([pu, pv], 1)
    ref = ops.klt_track(img1, img2, pts, 3, 2.0, (21, 21), 30, index_base=0, ctx=ctx)

    def cycle():
        c = _lib.Context(0)
        try:
            out = ops.klt_track(img1, img2, pts, 3, 2.0, (21, 21), 30, index_base=0, ctx=c)
            sel = ops.select_candidates(img1, pts[:100], keep=50, index_base=0, ctx=c)
            assert len(sel[0]) > 0
            for a, b in zip(out, ref):
                assert np.array_equal(a, b, equal_nan=True)
        finally:
            c.close()

    for _ in range(5):
        cycle()
    d0, h0 = device_used(), host_rss()
    for _ in range(40):
        cycle()
    d1, h1 = device_used(), host_rss()
    assert d1 - d0 <= (4 << 20), (d0, d1)          # nothing accumulates on the device
    assert h1 - h0 <= (64 << 20), (h0, h1)         # nor on the host (allocator slack allowed)


def test_runs_give_back_their_memory(ctx, tmp_path):
    spec = importlib.util.spec_from_file_location("run_dataset", os.path.join(ROOT, "scripts", "run_dataset.py"))
    rd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rd)
    n = 8
    rd.write_parking_tree(str(tmp_path), n + 3)

    def run():
        loader = ingest.dataLoaderParking(tmp_path, last_frame=n + 2)
        traj, info = pipeline.run_sequence(loader, num_frames=n, bootstrap_frames=(0, 2), MinNumLandmarks=50, threads=3)
        assert info["frames"] == n
        return traj.poses

    first = run()
    for _ in range(2):
        run()
    d0, h0 = device_used(), host_rss()
    for _ in range(12):
        assert np.array_equal(run(), first)        # and every run is the same run
    d1, h1 = device_used(), host_rss()
    assert d1 - d0 <= (4 << 20), (d0, d1)
    assert h1 - h0 <= (64 << 20), (h0, h1)


def test_a_pool_closed_early_stops_its_workers(ctx, tmp_path):
    """Closing a decode pool that still has work queued joins its threads and frees its slots."""
    import threading
    from oracle import ingest as oi
    rng = np.random.default_rng(54)
    paths = []
    for i in range(40):
        p = tmp_path / ("f%03d.png" % i)
        p.write_bytes(oi.png_encode(rng.integers(0, 256, (120, 160), dtype=np.uint8)))
        paths.append(str(p))
    for _ in range(3):
        fs = ingest.FrameStream(paths, threads=4, depth=6)
        next(fs)
        fs.close()
    import psutil
    t0, n0, d0 = threading.active_count(), psutil.Process().num_threads(), device_used()
    for _ in range(25):
        fs = ingest.FrameStream(paths, threads=4, depth=6)
        next(fs); next(fs)
        fs.close()
    assert threading.active_count() == t0
    assert psutil.Process().num_threads() <= n0 + 2     # the native workers (4 per pool) are joined at close
    assert device_used() - d0 <= (4 << 20)


def test_two_contexts_on_two_threads(ctx):
    """include/vo_hip.h: one vo_ctx per thread.  Two host threads, each with a context of its own, run the path's
    entry points at the same time (ctypes drops the GIL inside the calls) and get the bits a lone call gets."""
    import threading
    img1, img2, _ = synth.frame_pair(nv=240, nu=320, seed=55, shift=(0.9, -0.8))
    pu, pv = synth.keypoints(3000, nv=240, nu=320, seed=56)
    pts = np.stack([pu, pv], 1)
    rng = np.random.default_rng(57)
    Kc = np.array([[300.0, 0, 160], [0, 300.0, 120], [0, 0, 1]])
    X = np.column_stack([rng.uniform(-2, 2, 800), rng.uniform(-1.5, 1.5, 800), rng.uniform(4, 9, 800)])
    uvw = X @ Kc.T
    xy = uvw[:, :2] / uvw[:, 2:] + rng.normal(0, 0.3, (800, 2))
    xy[::5] += rng.uniform(-40, 40, (160, 2))

    def work(c):
        a = ops.klt_track(img1, img2, pts, 3, 2.0, (31, 31), 30, index_base=0, ctx=c)
        b = ops.select_candidates(img2, pts[:500], keep=80, index_base=0, want_all=True, ctx=c)
        r = ops.p3p_ransac(xy, X, Kc, num_hyp=600, seed=9, ctx=c)
        assert r["status"] == 0 and r["num_inliers"] > 400
        return list(a) + list(b) + [r["R"], r["t"], r["inliers"], np.array([r["best_hyp"], r["best_sol"], r["trials"]])]

    ref = work(ctx)
    results, errors = {}, []

    def runner(k):
        c = _lib.Context(0)
        try:
            for _ in range(6):
                out = work(c)
            results[k] = out
        except Exception as e:                                 # surfaced below: a thread's exception is not pytest's
            errors.append(e)
        finally:
            c.close()

    threads = [threading.Thread(target=runner, args=(k,)) for k in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for k in range(2):
        assert len(results[k]) == len(ref)
        for got, exp in zip(results[k], ref):
            assert np.array_equal(got, exp, equal_nan=True)
