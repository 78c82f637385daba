"""A second, independent restatement of the selection path's definition (oracle/vo_oracle.h N6-N8: min-eigen response,
regional maxima with sub-pixel refinement, the single-precision distance filter, selectUniform) in numpy -- whole-array
operations on padded / shifted copies, a sort per selectUniform round -- sharing no code or loop structure with
oracle/select.c, and oracle/select.c held to it BIT FOR BIT on three images (one of them KITTI-sized).  These stages
carry the bit-exact contract; the HIP kernels are checked against oracle/select.c on the GPU, and vo_select.hip takes
its weights from the same rule as oracle/select.c -- this file is what stands behind both.
Reference: modules/selectCandidateKeypoints.m:20-35.  PARITY UNPINNED all the same: detectMinEigenFeatures and
selectUniform are closed toolbox code."""
import math

import numpy as np
import pytest

from oracle import oracle
from visual_odometry_amd import synth


def gauss_weights(fs):
    """N6: w_k = round-half-away(4096 exp(-k^2 / (2 s^2)) / sum), s = fs / 3"""
    k = np.arange(fs, dtype=np.float64) - fs // 2
    s = fs / 3.0
    g = np.exp(-(k * k) / (2.0 * s * s))
    return np.floor(4096.0 * g / g.sum() + 0.5).astype(np.int64)


def response(img, fs=5):
    """N6, min-eigen: integer central differences with replicated borders, integer-weight separable Gaussian
    (u first, then v) in int64, metric = 0.5 ((A + C) - sqrt((A - C)^2 + 4 B^2)) 2^-26 in doubles, negatives -> 0."""
    I = np.pad(img.astype(np.int64), 1, mode="edge")
    du = I[1:-1, 2:] - I[1:-1, :-2]
    dv = I[2:, 1:-1] - I[:-2, 1:-1]
    w, r = gauss_weights(fs), fs // 2
    nv, nu = img.shape
    out = []
    for t in (du * du, du * dv, dv * dv):
        p = np.pad(t, ((0, 0), (r, r)), mode="edge")
        h = sum(w[k] * p[:, k:k + nu] for k in range(fs))
        p = np.pad(h, ((r, r), (0, 0)), mode="edge")
        out.append(sum(w[k] * p[k:k + nv, :] for k in range(fs)).astype(np.float64))
    A, B, C = out
    dif = A - C
    lam = 0.5 * ((A + C) - np.sqrt(dif * dif + 4.0 * (B * B)))
    return np.where(lam < 0.0, 0.0, lam) * 2.0 ** -26


def detect(resp, min_quality):
    """N7: metric > min_quality * max, off the 1-px border, (metric, -linear index) strictly greatest among the 8
    neighbours; parabola offsets clamped to +-0.5 where the curvature is negative; ascending linear index."""
    nv, nu = resp.shape
    c = resp[1:-1, 1:-1]
    ok = c > min_quality * resp.max()
    for dv in (-1, 0, 1):
        for du in (-1, 0, 1):
            if du == 0 and dv == 0:
                continue
            q = resp[1 + dv:nv - 1 + dv, 1 + du:nu - 1 + du]
            earlier = dv < 0 or (dv == 0 and du < 0)        # the neighbour's linear index is the smaller one
            ok &= (c > q) if earlier else (c >= q)
    v, u = np.nonzero(ok)
    v += 1; u += 1
    r = resp[v, u]
    rum, rup, rvm, rvp = resp[v, u - 1], resp[v, u + 1], resp[v - 1, u], resp[v + 1, u]
    cuu = (rup - 2.0 * r) + rum
    cvv = (rvp - 2.0 * r) + rvm
    with np.errstate(divide="ignore", invalid="ignore"):
        ou = np.where(cuu < 0, np.clip(-0.5 * (rup - rum) / cuu, -0.5, 0.5), 0.0)
        ov = np.where(cvv < 0, np.clip(-0.5 * (rvp - rvm) / cvv, -0.5, 0.5), 0.0)
    return (v * nu + u).astype(np.int32), (u + ou).astype(np.float32), (v + ov).astype(np.float32), r


def distance_keep(cu, cv, eu, ev, mind):
    """N8 / modules/selectCandidateKeypoints.m:23-29: keep a corner unless an existing point has
    sqrt(single(dx^2 + dy^2)) < MinDistance, everything in single precision (the distance matrix of pdist2)."""
    keep = np.ones(len(cu), bool)
    if len(eu) == 0:
        return keep
    eu = eu.astype(np.float32); ev = ev.astype(np.float32)
    md = np.float32(mind)
    for i0 in range(0, len(cu), 256):
        dx = cu[i0:i0 + 256, None] - eu[None, :]
        dy = cv[i0:i0 + 256, None] - ev[None, :]
        d = np.sqrt(dx * dx + dy * dy)                     # float32 throughout
        assert d.dtype == np.float32
        keep[i0:i0 + 256] = ~(d < md).any(axis=1)
    return keep


def select_uniform(cu, cv, metric, keep, n_keep, nu, nv, u_is_rows=False):
    """N8 selectUniform: rounds of 'strongest corner of every grid cell' (ties: the lower index) on the corners not yet
    taken, grid h = max(floor(sqrt(n / aspect)), 1) by w = max(floor(h aspect), 1) for the n points still wanted."""
    sel = np.zeros(len(cu), bool)
    H, W = (nu, nv) if u_is_rows else (nv, nu)
    while sel.sum() < n_keep and (keep & ~sel).any():
        want = n_keep - int(sel.sum())
        h = min(max(math.isqrt((want * H) // W), 1), want)
        w = max((h * W) // H, 1)
        if h * w > want:
            w = want // h
        cells_u, cells_v = (h, w) if u_is_rows else (w, h)
        idx = np.nonzero(keep & ~sel)[0]
        bu = np.clip(np.floor((cu[idx].astype(np.float64) * cells_u) / nu).astype(np.int64), 0, cells_u - 1)
        bv = np.clip(np.floor((cv[idx].astype(np.float64) * cells_v) / nv).astype(np.int64), 0, cells_v - 1)
        cell = bv * cells_u + bu
        order = np.lexsort((idx, -metric[idx], cell))       # by cell, then metric descending, then index ascending
        first = np.ones(len(order), bool)
        first[1:] = cell[order][1:] != cell[order][:-1]
        sel[idx[order[first]]] = True
    return sel


IMAGES = [("golden-size", 96, 128, 3), ("qvga", 240, 320, 5), ("kitti", 376, 1241, 7)]


@pytest.mark.parametrize("name,nv,nu,seed", IMAGES)
def test_response_and_detection_bit_for_bit(name, nv, nu, seed):
    img = synth.frame_pair(nv=nv, nu=nu, seed=seed)[0]
    for fs in (5, 3, 7):
        assert np.array_equal(gauss_weights(fs), oracle.gauss_weights(fs))
    for fs in ((5, 3) if nu < 1000 else (5,)):
        mine = response(img, fs)
        theirs = oracle.corner_response(img, 0, fs)
        assert np.array_equal(mine, theirs), f"response differs at filter size {fs}"
    resp = oracle.corner_response(img, 0, 5)
    for q in (0.001, 0.01):
        lin, lu, lv, met = detect(resp, q)
        olin, olu, olv, omet = oracle.detect_corners(resp, q)
        assert len(lin) > 50
        assert np.array_equal(lin, olin) and np.array_equal(met, omet)
        assert np.array_equal(lu, olu) and np.array_equal(lv, olv)       # float32 locations, bit for bit


@pytest.mark.parametrize("name,nv,nu,seed", IMAGES)
@pytest.mark.parametrize("keep,n_exist", [(100, 0), (100, 400), (2000, 3000)])
def test_whole_selection_bit_for_bit(name, nv, nu, seed, keep, n_exist):
    img = synth.frame_pair(nv=nv, nu=nu, seed=seed)[0]
    eu, ev = synth.keypoints(max(n_exist, 1), nv=nv, nu=nu, seed=seed + 100, border=2)
    eu, ev = eu[:n_exist].astype(np.float32), ev[:n_exist].astype(np.float32)
    lin, lu, lv, met = detect(response(img), 0.001)
    kept = distance_keep(lu, lv, eu, ev, 7.0)
    for col_major in (False, True):
        sel = select_uniform(lu, lv, met, kept, keep, nu, nv, col_major)
        ou, ov, olin, omet = oracle.select_candidates(img, eu, ev, min_distance=7.0, min_quality=0.001, keep=keep, col_major=col_major)
        assert np.array_equal(lin[sel], olin), (name, keep, n_exist, col_major)
        assert np.array_equal(lu[sel], ou) and np.array_equal(lv[sel], ov) and np.array_equal(met[sel], omet)
        assert sel.sum() <= keep and (sel.sum() > 0 or n_exist * 150 > nu * nv)   # (a crowded small image keeps nothing)


def test_ties_and_the_strict_distance_rule():
    """plateaus (equal responses side by side: the lower linear index is the corner), a corner exactly at MinDistance
    stays (strict <), one just inside goes -- on a hand-made response, both restatements"""
    resp = np.zeros((12, 16))
    resp[3, 4] = resp[3, 5] = 2.0          # a two-pixel plateau: (3, 4) wins
    resp[8, 10] = 1.0
    resp[8, 3] = resp[9, 3] = 1.5          # vertical plateau: (8, 3) wins
    lin, lu, lv, met = detect(resp, 0.01)
    olin, olu, olv, omet = oracle.detect_corners(resp, 0.01)
    assert np.array_equal(lin, olin) and list(lin) == [3 * 16 + 4, 8 * 16 + 3, 8 * 16 + 10]
    assert np.array_equal(lu, olu) and np.array_equal(lv, olv)
    cu = np.array([10.0, 20.0, 30.0], np.float32); cv = np.array([5.0, 5.0, 5.0], np.float32)
    eu = np.array([17.0, 37.0 - 2 ** -18], np.float32); ev = np.array([5.0, 5.0], np.float32)
    keep = distance_keep(cu, cv, eu, ev, 7.0)
    assert list(keep) == [True, False, False]       # 10 is exactly 7 from 17: kept (strict <); 20 is 3 away; 30 is 7 - 2^-18 away
    from oracle.oracle import lib, _p
    import ctypes as C
    k2 = np.zeros(3, np.uint8)
    lib().vo_or_distance_filter(_p(cu), _p(cv), 3, _p(eu), _p(ev), 2, C.c_float(7.0), _p(k2))
    assert list(k2.astype(bool)) == list(keep)
