"""A second, independent restatement of the tracker's definition (oracle/vo_oracle.h N1-N5) -- written from Bouguet's
"Pyramidal Implementation of the Lucas Kanade Feature Tracker" with numpy / scipy.ndimage building blocks (separable
correlation with replicate borders for the pyramid, map_coordinates for the bilinear samples, a 3x3 correlation for
Sobel, eigvalsh / solve for the 2x2 system), sharing no code or loop structure with oracle/klt.c -- and oracle/klt.c
held to it at 1e-9 px on the golden 96 x 128 pair.  k_lk (the dominant kernel) is checked against oracle/klt.c on the
GPU; this file is what stands behind oracle/klt.c, the way harris_numpy stands behind the Harris scorer.
PARITY UNPINNED all the same: vision.PointTracker's internals are closed."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import oracle
from tests.test_golden import KLT_CFG

MIN_EIG = 1e-3          # vo_oracle.h VO_MIN_EIG
W5 = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
SOBEL_U = np.array([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0     # rows = v, columns = u
SOBEL_V = SOBEL_U.T


def pyramid(img, levels):
    """N1: [1 4 6 4 1]/16 along u then v with replicate borders, every second sample, size ceil(n/2)."""
    out = [img.astype(np.float64)]
    for _ in range(1, levels):
        a = ndimage.correlate1d(out[-1], W5, axis=1, mode="nearest")[:, ::2]
        out.append(ndimage.correlate1d(a, W5, axis=0, mode="nearest")[::2, :])
    return out


def window(I, cu, cv, ru, rv):
    """N3: bilinear samples of I on the integer-offset grid around (cu, cv); replicate outside the image."""
    du = np.arange(-ru, ru + 1, dtype=np.float64); dv = np.arange(-rv, rv + 1, dtype=np.float64)
    vv, uu = np.meshgrid(cv + dv, cu + du, indexing="ij")
    return ndimage.map_coordinates(I, [vv, uu], order=1, mode="nearest")


def track_one(A, B, p, levels, wu, wv, max_iters, eps):
    """N4, one direction.  -> (q, lost, flat0, ssd, iterations)"""
    ru, rv = (wu - 1) // 2, (wv - 1) // 2
    nv0, nu0 = A[0].shape
    g = np.zeros(2)
    iters = 0
    flat0 = False
    T0 = None
    inside = lambda c, s: 0.0 <= c[0] * s <= nu0 - 1 and 0.0 <= c[1] * s <= nv0 - 1
    for l in range(levels - 1, -1, -1):
        s = float(1 << l)
        pl = np.asarray(p) / s
        E = window(A[l], pl[0], pl[1], ru + 1, rv + 1)                # the window grown by one pixel
        T = E[1:-1, 1:-1]
        Tu = ndimage.correlate(E, SOBEL_U, mode="nearest")[1:-1, 1:-1]  # Sobel/8 of the sampled patch (interior only,
        Tv = ndimage.correlate(E, SOBEL_V, mode="nearest")[1:-1, 1:-1]  # so the border mode never enters)
        G = np.array([[np.sum(Tu * Tu), np.sum(Tu * Tv)], [np.sum(Tu * Tv), np.sum(Tv * Tv)]])
        d = np.zeros(2)
        if not (np.linalg.eigvalsh(G)[0] / (wu * wv) >= MIN_EIG):
            flat0 = flat0 or l == 0
        else:
            for _ in range(max_iters):
                c = pl + g + d
                if not inside(c, s):
                    return np.asarray(p) + g, True, flat0, 0.0, iters      # lost: the caller ignores q
                J = window(B[l], c[0], c[1], ru, rv)
                e = T - J
                delta = np.linalg.solve(G, np.array([np.sum(e * Tu), np.sum(e * Tv)]))
                iters += 1
                d = d + delta
                if delta @ delta < eps * eps:
                    break
        g = 2.0 * (g + d) if l > 0 else g + d
        if l == 0:
            T0 = T
    q = np.asarray(p) + g
    if not inside(q, 1.0):
        return q, True, flat0, 0.0, iters
    e = T0 - window(B[0], q[0], q[1], ru, rv)                          # N5 residual at level 0
    return q, False, flat0, float(np.sum(e * e)), iters


def klt_second(img1, img2, pu, pv, levels, wu, wv, max_iters, max_bidir, eps=0.01):
    A, B = pyramid(img1, levels), pyramid(img2, levels)
    n = len(pu)
    q = np.zeros((n, 2)); valid = np.zeros(n, bool); score = np.zeros(n); fb = np.full(n, np.nan); it = np.zeros(n, np.int32)
    for i in range(n):
        p = np.array([pu[i], pv[i]])
        qi, lost, flat0, ssd, k = track_one(A, B, p, levels, wu, wv, max_iters, eps)
        ok = not lost and not flat0
        if not lost and max_bidir >= 0:                                  # N5: backward track from q
            pb, lost_b, _, _, kb = track_one(B, A, qi, levels, wu, wv, max_iters, eps)
            k += kb
            if lost_b:
                ok = False
            else:
                fb[i] = np.hypot(*(p - pb))
                ok = ok and fb[i] <= max_bidir
        q[i] = qi; valid[i] = ok; it[i] = k
        score[i] = 0.0 if lost else min(max(1.0 - np.sqrt(ssd / (wu * wv)) / 255.0, 0.0), 1.0)
    return q[:, 0], q[:, 1], valid, score, fb, it


@pytest.mark.parametrize("tag", ["s", "l", "m"])
def test_klt_c_oracle_equals_the_second_restatement(golden, tag):
    g, kw = golden, KLT_CFG[tag]
    o = oracle.klt_track(g["klt_img1"], g["klt_img2"], g["klt_pu"], g["klt_pv"], **kw)
    s = klt_second(g["klt_img1"], g["klt_img2"], g["klt_pu"], g["klt_pv"], kw["levels"], kw["wu"], kw["wv"], kw["max_iters"],
                   kw["max_bidir"])
    np.testing.assert_array_equal(o[2], s[2])                            # validity
    np.testing.assert_array_equal(o[5], s[5])                            # executed iterations, both directions
    lost = np.isnan(o[4]) if kw["max_bidir"] >= 0 else np.zeros(len(o[0]), bool)
    keep = ~lost                                                         # a lost point's position is not defined
    assert keep.sum() >= 40
    assert np.abs(o[0] - s[0])[keep].max() < 1e-9 and np.abs(o[1] - s[1])[keep].max() < 1e-9
    np.testing.assert_allclose(o[3][keep], s[3][keep], atol=1e-12)       # score
    both = ~np.isnan(o[4])
    assert np.array_equal(both, ~np.isnan(s[4])) and (not both.any() or np.abs(o[4] - s[4])[both].max() < 1e-9)
    assert 0 < o[2].sum()


def test_pyramid_second_restatement(golden):
    img = golden["klt_img1"]
    nv, nu = img.shape
    mine = pyramid(img, 4)
    pyr = np.zeros(sum(m.size for m in mine))
    nus = np.zeros(8, np.int32); nvs = np.zeros(8, np.int32); offs = np.zeros(8, np.int64)
    import ctypes as C
    from oracle.oracle import lib, _p
    lib().vo_or_build_pyramid(_p(np.ascontiguousarray(img)), nu, nv, 4, _p(pyr))
    off = 0
    for m in mine:
        np.testing.assert_allclose(pyr[off:off + m.size].reshape(m.shape), m, atol=1e-11)
        off += m.size
