"""N10b (oracle/five_point.c): Nister's five-point solver and the MSAC loop around it, on the CPU.
The solver against ground truth (general and PLANAR scenes: the case the eight-point algorithm cannot do), against the
epipolar constraints themselves, and against an independent numpy restatement (tests/five_point_np.py); the loop on a
planar two-view scene with outliers."""
import numpy as np
import pytest

from oracle import oracle
from five_point_np import five_point_np


def rot(w):
    th = np.linalg.norm(w); k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def scene(rng, n, planar):
    R = rot(rng.normal(size=3) * 0.2); t = rng.normal(size=3); t /= np.linalg.norm(t)
    X = rng.uniform(-1, 1, (n, 3)) + [0, 0, 5]
    if planar:
        X[:, 2] = 5 + 0.3 * X[:, 0] - 0.2 * X[:, 1]
    Y = (R @ X.T).T + t
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:], E / np.linalg.norm(E) * np.sqrt(2)


def closest(Es, Et):
    return min((min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es), default=9.0)


@pytest.mark.parametrize("planar", [False, True])
def test_five_point_recovers_the_true_matrix(planar):
    """The true matrix is among the solutions of (nearly) every sample.  Not of every one: E = x X + y Y + z Z + W fixes
    the coefficient of the fourth null vector to 1, so a sample whose true matrix has a small component along W is
    ill-conditioned -- in any implementation (the numpy restatement misses 0.3 % / 2-5 % of these samples, depending on
    the basis it is given) -- and coplanar points make that more frequent.  The MSAC loop draws another sample."""
    rng = np.random.default_rng(7 + planar)
    errs = []
    for _ in range(600):
        x0, x1, Et = scene(rng, 5, planar)
        Es = oracle.five_point(x0, x1)
        assert 1 <= len(Es) <= 10
        errs.append(closest(Es, Et))
        for E in Es:      # every solution lies on the essential manifold
            s = np.linalg.svd(E, compute_uv=False)
            assert abs(s[0] - 1) < 1e-9 and abs(s[1] - 1) < 1e-9 and s[2] < 1e-9
    errs = np.array(errs)
    if planar:
        assert np.mean(errs < 1e-6) > 0.85 and np.mean(errs < 1e-3) > 0.93, (np.mean(errs < 1e-6), np.mean(errs < 1e-3))
    else:
        assert np.mean(errs < 1e-6) > 0.98 and np.mean(errs < 1e-3) > 0.995, (np.mean(errs < 1e-6), np.mean(errs < 1e-3))
    assert np.median(errs) < 1e-10


def test_five_point_solutions_equal_the_numpy_restatement():
    """Solution by solution against tests/five_point_np.py on the same samples: the two agree on the well-conditioned
    solutions (most of them to 1e-9), and where one of them has a solution the other one has one nearby."""
    rng = np.random.default_rng(3)
    missing = extra = total = close = 0
    for trial in range(200):
        x0, x1, Et = scene(rng, 5, trial % 2 == 1)
        Es = oracle.five_point(x0, x1)
        q0 = np.c_[x0, np.ones(5)]; q1 = np.c_[x1, np.ones(5)]
        ref = five_point_np(x0[:, 0], x0[:, 1], x1[:, 0], x1[:, 1])
        total += len(ref)
        d = [closest(Es, Er) for Er in ref]
        close += sum(x < 1e-9 for x in d)
        missing += sum(x > 1e-3 for x in d)                       # a solution the oracle did not find
        extra += sum(closest(ref, E) > 1e-3 for E in Es)
        for E in Es:                                              # through the five points (to the accuracy of the root)
            assert np.abs(np.einsum("ni,ij,nj->n", q1, E, q0)).max() < 1e-2
    assert total > 600 and close > 0.7 * total and missing <= 0.03 * total and extra <= 0.03 * total, (total, close, missing, extra)


def test_solutions_come_in_ascending_order_and_samples_are_distinct():
    rng = np.random.default_rng(11)
    for m in (5, 6, 9, 100, 4000):
        for h in range(50):
            idx = oracle.philox_sample5(99, h, m)
            assert len(set(idx.tolist())) == 5 and idx.min() >= 0 and idx.max() < m
    # the first five picks of the eight-point sampler
    assert np.array_equal(oracle.philox_sample5(5, 17, 1000), oracle.philox_sample8(5, 17, 1000)[:5])


@pytest.mark.parametrize("planar", [False, True])
def test_msac_loop_on_a_two_view_scene_with_outliers(planar):
    rng = np.random.default_rng(21 + planar)
    K = np.array([[700.0, 0, 600], [0, 700.0, 180], [0, 0, 1]])
    x0, x1, Et = scene(rng, 400, planar)
    p0 = x0 * 700 + [600, 180]; p1 = x1 * 700 + [600, 180]
    p1 += rng.normal(0, 0.05, p1.shape)
    bad = rng.choice(400, 100, replace=False)
    p1[bad] += rng.uniform(-40, 40, (100, 2))
    out = oracle.essential_ransac5(p0, p1, K, num_hyp=300, max_distance=0.09, confidence=0.995, seed=5, want_all=True)
    assert out["status"] == 0 and out["trials"] < 300
    good = np.setdiff1d(np.arange(400), bad)
    assert out["inliers"][good].mean() > 0.9 and out["inliers"][bad].mean() < 0.1
    # (coplanar points are consistent with TWO essential matrices -- the two decompositions of their homography -- so the
    # winner of a planar scene need not be the one the scene was made with; the inlier set is the same)
    if not planar:
        assert closest([out["E"]], Et) < 5e-3
    # the record is the first slot, in order, that no earlier slot beats (strictly smaller cost wins)
    cost = out["costs"][: 10 * out["trials"]].astype(np.float64)
    ok = out["counts"][: 10 * out["trials"]] >= 5
    cost[~ok] = np.inf
    assert out["best_hyp"] == int(np.argmin(cost))
