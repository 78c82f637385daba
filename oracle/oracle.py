"""ctypes binding of the CPU parity oracle (oracle/libvo_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  Parity is UNPINNED by the reference (pure MATLAB over closed-source
toolboxes, no golden vectors): see oracle/vo_oracle.h for the normative definitions.

All functions work in the INTERNAL image convention (u = fast axis, v = slow axis, 0-based).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "libvo_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvo_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.vo_or_angle_deg.restype = C.c_double
        _LIB.vo_or_pyramid_layout.restype = C.c_long
    return _LIB


def _bind(path):
    L = C.CDLL(path)
    L.vo_or_angle_deg.restype = C.c_double
    L.vo_or_pyramid_layout.restype = C.c_long
    return L


def use_timing_build():
    """bench.py's cpu_baseline leg only: switch this process to an optimised build of the same sources -- -O3
    -march=native compiled on THIS machine (BASELINE.md's flags; needs gcc), else the x86-64-v3 build that travels
    with the tree, else the -O2 parity build stays.  Returns a description of what is loaded."""
    global _LIB
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="vo_oracle_native_"), "libvo_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "native", "OUT=" + out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        _LIB = _bind(out)
        return "-O3 -march=native, built on this host"
    except Exception:
        pass
    v3 = os.path.join(_HERE, "libvo_oracle_v3.so")
    if os.path.exists(v3):
        # the instruction set may be missing on this host: try it in a child first
        ok = subprocess.call([sys.executable, "-c", "import ctypes,sys; L=ctypes.CDLL(sys.argv[1]); import array; w=(ctypes.c_int32*16)(); L.vo_or_gauss_weights(5, w)", v3],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0
        if ok:
            _LIB = _bind(v3)
            return "-O3 -march=x86-64-v3 (prebuilt)"
    lib()
    return "-O2 (the parity build)"


def set_threads(n):
    """threads of the independent loops in selection and triangulation (the tracker and the pose step take theirs
    per call); results do not depend on it"""
    lib().vo_or_set_threads(int(n))


class KltOpts(C.Structure):
    _fields_ = [("levels", C.c_int), ("max_bidir", C.c_double), ("wu", C.c_int), ("wv", C.c_int),
                ("max_iters", C.c_int), ("eps", C.c_double), ("threads", C.c_int)]


class RansacOpts(C.Structure):
    _fields_ = [("num_hyp", C.c_int), ("max_reproj", C.c_double), ("confidence", C.c_double),
                ("seed", C.c_uint64), ("mode", C.c_int), ("min_inliers", C.c_int), ("threads", C.c_int)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def pyramid_layout(nu, nv, levels):
    nus = (C.c_int * 8)(); nvs = (C.c_int * 8)(); offs = (C.c_long * 8)()
    tot = lib().vo_or_pyramid_layout(nu, nv, levels, nus, nvs, offs)
    return tot, list(nus)[:levels], list(nvs)[:levels], list(offs)[:levels]


def build_pyramid(img, levels):
    """img: 2-D uint8, memory order [v][u].  Returns list of float64 level arrays."""
    img = _c(img, np.uint8)
    nv, nu = img.shape
    tot, nus, nvs, offs = pyramid_layout(nu, nv, levels)
    pyr = np.empty(tot, np.float64)
    lib().vo_or_build_pyramid(_p(img), nu, nv, levels, _p(pyr))
    return [pyr[offs[l]:offs[l] + nus[l] * nvs[l]].reshape(nvs[l], nus[l]) for l in range(levels)]


def gradients(I):
    I = _c(I, np.float64)
    nv, nu = I.shape
    gu = np.empty_like(I); gv = np.empty_like(I)
    lib().vo_or_gradients(_p(I), nu, nv, _p(gu), _p(gv))
    return gu, gv


def klt_track(img1, img2, pu, pv, levels=4, max_bidir=2.0, wu=41, wv=41, max_iters=50, eps=0.01, threads=1):
    img1 = _c(img1, np.uint8); img2 = _c(img2, np.uint8)
    nv, nu = img1.shape
    pu = _c(pu, np.float64); pv = _c(pv, np.float64)
    n = pu.shape[0]
    qu = np.zeros(n); qv = np.zeros(n); valid = np.zeros(n, np.uint8); score = np.zeros(n)
    fb = np.zeros(n); iters = np.zeros(n, np.int32)
    o = KltOpts(levels, max_bidir, wu, wv, max_iters, eps, threads)
    rc = lib().vo_or_klt_track(_p(img1), _p(img2), nu, nv, _p(pu), _p(pv), n, C.byref(o), _p(qu), _p(qv),
                               _p(valid), _p(score), _p(fb), _p(iters))
    if rc != 0:
        raise ValueError("vo_or_klt_track: bad options")
    return qu, qv, valid.astype(bool), score, fb, iters


def gauss_weights(fs):
    w = np.zeros(fs, np.int32)
    lib().vo_or_gauss_weights(fs, _p(w))
    return w


def corner_response(img, scorer=0, filter_size=5, kappa=0.08):
    img = _c(img, np.uint8)
    nv, nu = img.shape
    resp = np.empty((nv, nu), np.float64)
    lib().vo_or_corner_response(_p(img), nu, nv, scorer, filter_size, C.c_double(kappa), _p(resp))
    return resp


def detect_corners(resp, min_quality):
    resp = _c(resp, np.float64)
    nv, nu = resp.shape
    cap = nu * nv // 4 + 16
    lin = np.zeros(cap, np.int32); lu = np.zeros(cap, np.float32); lv = np.zeros(cap, np.float32)
    met = np.zeros(cap, np.float64)
    m = lib().vo_or_detect_corners(_p(resp), nu, nv, C.c_double(min_quality), cap, _p(lin), _p(lu), _p(lv), _p(met))
    return lin[:m].copy(), lu[:m].copy(), lv[:m].copy(), met[:m].copy()


def select_candidates(img, eu, ev, min_distance=7.0, min_quality=0.001, filter_size=5, keep=100, scorer=0,
                      kappa=0.08, col_major=False):
    img = _c(img, np.uint8)
    nv, nu = img.shape
    eu = _c(eu, np.float32); ev = _c(ev, np.float32)
    ou = np.zeros(keep, np.float32); ov = np.zeros(keep, np.float32); ol = np.zeros(keep, np.int32)
    om = np.zeros(keep, np.float64)
    m = lib().vo_or_select_candidates(_p(img), nu, nv, _p(eu), _p(ev), eu.shape[0], C.c_float(min_distance),
                                      C.c_double(min_quality), filter_size, keep, scorer, C.c_double(kappa),
                                      int(col_major), _p(ou), _p(ov), _p(ol), _p(om))
    return ou[:m].copy(), ov[:m].copy(), ol[:m].copy(), om[:m].copy()


def select_keypoints_greedy(scores, num, r):
    scores = _c(scores, np.float64)
    nv, nu = scores.shape
    ou = np.zeros(num, np.int32); ov = np.zeros(num, np.int32)
    lib().vo_or_select_keypoints_greedy(_p(scores), nu, nv, num, r, _p(ou), _p(ov))
    return ou, ov


def camera_matrix(pose43, K):
    pose43 = _c(pose43, np.float64); K = _c(K, np.float64)
    cam = np.zeros((4, 3))
    lib().vo_or_camera_matrix(_p(pose43), _p(K), _p(cam))
    return cam


def angle_deg(X, pose0, pose1):
    X = _c(X, np.float64); pose0 = _c(pose0, np.float64); pose1 = _c(pose1, np.float64)
    return lib().vo_or_angle_deg(_p(X), _p(pose0), _p(pose1))


def in_front(cam43, X):
    cam43 = _c(cam43, np.float64); X = _c(X, np.float64).reshape(-1, 3)
    v = np.zeros(X.shape[0], np.uint8)
    lib().vo_or_in_front(_p(cam43), _p(X), X.shape[0], _p(v))
    return v.astype(bool)


def triangulate(xy0, xy1, poses0, pose1, K):
    xy0 = _c(xy0, np.float64).reshape(-1, 2); xy1 = _c(xy1, np.float64).reshape(-1, 2)
    n = xy0.shape[0]
    poses0 = _c(poses0, np.float64).reshape(n, 12); pose1 = _c(pose1, np.float64); K = _c(K, np.float64)
    X = np.zeros((n, 3)); rep = np.zeros(n); ang = np.zeros(n); val = np.zeros(n, np.uint8)
    lib().vo_or_triangulate(_p(xy0), _p(xy1), _p(poses0), _p(pose1), _p(K), n, _p(X), _p(rep), _p(ang), _p(val))
    return X, rep, ang, val.astype(bool)


def triangulate_cam(xy0, xy1, cam0, cam1):
    """the toolbox call `triangulate(p1, p2, camMatrix1, camMatrix2)` (modules/VisualOdometry.m:97-101): 4x3 matrices given"""
    xy0 = _c(xy0, np.float64).reshape(-1, 2); xy1 = _c(xy1, np.float64).reshape(-1, 2)
    n = xy0.shape[0]
    cam0 = _c(cam0, np.float64).reshape(4, 3); cam1 = _c(cam1, np.float64).reshape(4, 3)
    X = np.zeros((n, 3)); rep = np.zeros(n); val = np.zeros(n, np.uint8)
    lib().vo_or_triangulate_cam(_p(xy0), _p(xy1), _p(cam0), _p(cam1), n, _p(X), _p(rep), _p(val))
    return X, rep, val.astype(bool)


def reproj_error(xy, X, pose43, K):
    xy = _c(xy, np.float64).reshape(-1, 2); X = _c(X, np.float64).reshape(-1, 3)
    pose43 = _c(pose43, np.float64); K = _c(K, np.float64)
    err = np.zeros(xy.shape[0])
    lib().vo_or_reproj_error(_p(xy), _p(X), xy.shape[0], _p(pose43), _p(K), _p(err))
    return err


def solve_quartic(factors):
    f = _c(factors, np.float64)
    out = np.zeros((4, 2))
    lib().vo_or_solve_quartic(_p(f), _p(out))
    return out[:, 0] + 1j * out[:, 1]


def p3p(P, f):
    P = _c(P, np.float64).reshape(3, 3); f = _c(f, np.float64).reshape(3, 3)
    poses = np.zeros((4, 12))
    lib().vo_or_p3p(_p(P), _p(f), _p(poses))
    return poses


def philox_sample3(seed, hyp, m):
    idx = np.zeros(3, np.int32)
    lib().vo_or_philox_sample3(C.c_uint64(seed), C.c_uint32(hyp), C.c_uint32(m), _p(idx))
    return idx


def p3p_ransac(xy, X, K, num_hyp=2000, max_reproj=2.0, confidence=0.0, seed=4095, mode=0, min_inliers=0,
               sample_idx=None, threads=1, want_all=False):
    xy = _c(xy, np.float64).reshape(-1, 2); X = _c(X, np.float64).reshape(-1, 3); K = _c(K, np.float64)
    m = xy.shape[0]
    o = RansacOpts(num_hyp, max_reproj, confidence, seed, mode, min_inliers, threads)
    R = np.zeros((3, 3)); t = np.zeros(3); inl = np.zeros(max(m, 1), np.uint8)
    bh = C.c_int(); bs = C.c_int(); used = C.c_int()
    si = None if sample_idx is None else _c(sample_idx, np.int32)
    cnt = np.zeros((num_hyp, 4), np.int32); cost = np.zeros((num_hyp, 4), np.uint64)
    poses = np.zeros((num_hyp, 4, 12))
    st = lib().vo_or_p3p_ransac(_p(xy), _p(X), m, _p(K), C.byref(o), None if si is None else _p(si), _p(R), _p(t),
                                _p(inl), C.byref(bh), C.byref(bs), C.byref(used), _p(cnt), _p(cost), _p(poses))
    out = dict(status=st, R=R, t=t, inliers=inl[:m].astype(bool), best_hyp=bh.value, best_sol=bs.value,
               trials=used.value)
    if want_all:
        out.update(counts=cnt, costs=cost, poses=poses)
    return out


def philox_sample8(seed, hyp, m):
    idx = np.zeros(8, np.int32)
    lib().vo_or_philox_sample8(C.c_uint64(seed), C.c_uint32(hyp), C.c_uint32(m), _p(idx))
    return idx


def eight_point(x0, x1):
    """x0, x1: 8 x 2 K^-1-normalised points -> (ok, E) with x1' E x0 = 0, singular values (1, 1, 0)."""
    x0 = _c(x0, np.float64).reshape(8, 2); x1 = _c(x1, np.float64).reshape(8, 2)
    a0 = np.ascontiguousarray(x0[:, 0]); b0 = np.ascontiguousarray(x0[:, 1])
    a1 = np.ascontiguousarray(x1[:, 0]); b1 = np.ascontiguousarray(x1[:, 1])
    E = np.zeros((3, 3))
    ok = lib().vo_or_eight_point(_p(a0), _p(b0), _p(a1), _p(b1), _p(E))
    return bool(ok), E


def essential_ransac(p0, p1, K, num_hyp=2000, max_distance=0.09, confidence=0.995, seed=1023, min_inliers=8,
                     sample_idx=None, threads=1, want_all=False):
    """modules/bootstrap.m:76-79 (N10, oracle/essential.c)."""
    p0 = _c(p0, np.float64).reshape(-1, 2); p1 = _c(p1, np.float64).reshape(-1, 2); K = _c(K, np.float64)
    m = p0.shape[0]
    o = RansacOpts(num_hyp, max_distance, confidence, seed, 1, min_inliers, threads)
    E = np.zeros((3, 3)); inl = np.zeros(max(m, 1), np.uint8)
    bh = C.c_int(); used = C.c_int()
    si = None if sample_idx is None else _c(sample_idx, np.int32)
    cnt = np.zeros(num_hyp, np.int32); cost = np.zeros(num_hyp, np.uint64); Es = np.zeros((num_hyp, 3, 3))
    st = lib().vo_or_essential_ransac(_p(p0), _p(p1), m, _p(K), C.byref(o), None if si is None else _p(si), _p(E), _p(inl),
                                      C.byref(bh), C.byref(used), _p(cnt), _p(cost), _p(Es))
    out = dict(status=st, E=E, inliers=inl[:m].astype(bool), best_hyp=bh.value, trials=used.value)
    if want_all:
        out.update(counts=cnt, costs=cost, Es=Es)
    return out


def philox_sample5(seed, hyp, m):
    idx = np.zeros(5, np.int32)
    lib().vo_or_philox_sample5(C.c_uint64(seed), C.c_uint32(hyp), C.c_uint32(m), _p(idx))
    return idx


def five_point(x0, x1):
    """x0, x1: 5 x 2 K^-1-normalised points -> n x 3 x 3 matrices E (x1' E x0 = 0, singular values (1, 1, 0)), n <= 10,
    ascending in the root of Nister's tenth-degree polynomial (N10b, oracle/five_point.c)."""
    x0 = _c(x0, np.float64).reshape(5, 2); x1 = _c(x1, np.float64).reshape(5, 2)
    a0 = np.ascontiguousarray(x0[:, 0]); b0 = np.ascontiguousarray(x0[:, 1])
    a1 = np.ascontiguousarray(x1[:, 0]); b1 = np.ascontiguousarray(x1[:, 1])
    Es = np.zeros((10, 3, 3))
    n = lib().vo_or_five_point(_p(a0), _p(b0), _p(a1), _p(b1), _p(Es))
    return Es[:n]


def essential_ransac5(p0, p1, K, num_hyp=2000, max_distance=0.09, confidence=0.995, seed=1023, min_inliers=5,
                      sample_idx=None, threads=1, want_all=False):
    """modules/bootstrap.m:76-79 around the five-point solver (N10b, oracle/five_point.c); the per-hypothesis outputs
    have 10 slots per sample and best_hyp is a slot index (10 h + s)."""
    p0 = _c(p0, np.float64).reshape(-1, 2); p1 = _c(p1, np.float64).reshape(-1, 2); K = _c(K, np.float64)
    m = p0.shape[0]
    o = RansacOpts(num_hyp, max_distance, confidence, seed, 1, min_inliers, threads)
    E = np.zeros((3, 3)); inl = np.zeros(max(m, 1), np.uint8)
    bh = C.c_int(); used = C.c_int()
    si = None if sample_idx is None else _c(sample_idx, np.int32)
    cnt = np.zeros(10 * num_hyp, np.int32); cost = np.zeros(10 * num_hyp, np.uint64); Es = np.zeros((10 * num_hyp, 3, 3))
    st = lib().vo_or_essential_ransac5(_p(p0), _p(p1), m, _p(K), C.byref(o), None if si is None else _p(si), _p(E), _p(inl),
                                       C.byref(bh), C.byref(used), _p(cnt), _p(cost), _p(Es))
    out = dict(status=st, E=E, inliers=inl[:m].astype(bool), best_hyp=bh.value, trials=used.value)
    if want_all:
        out.update(counts=cnt, costs=cost, Es=Es)
    return out
