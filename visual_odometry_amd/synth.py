"""Seeded synthetic inputs of the shapes SURVEY.md section 8(d) names (no dataset ships with the
reference: data/.gitkeep only).  Pure numpy/scipy, host side; used by bench.py and the tests.

Images are returned as numpy uint8 arrays in memory order [v][u] (see include/vo_hip.h); for a
"numpy" image that is [y][x]; a MATLAB column-major H x W image is the same bytes seen as [x][y].
"""
import numpy as np
from scipy import ndimage

KITTI_K = np.array([[7.188560000000e+02, 0, 6.071928000000e+02],
                    [0, 7.188560000000e+02, 1.852157000000e+02],
                    [0, 0, 1]])  # utils/dataLoaderKitti.m:50-52


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def texture(nv, nu, seed=1023, sigma=2.0, margin=32):
    """Band-limited texture (float64, range [16,240]) with a margin for warping."""
    r = _rng(seed)
    t = r.integers(0, 256, size=(nv + 2 * margin, nu + 2 * margin)).astype(np.float64)
    t = ndimage.gaussian_filter(t, sigma, mode="wrap")
    t = (t - t.min()) / (t.max() - t.min())
    return 16.0 + 224.0 * t


def frame_pair(nv=376, nu=1241, seed=1023, shift=(3.3, -2.6), zoom=0.0, noise=0.0, margin=32):
    """Two uint8 frames: frame2(p + shift + zoom*(p-c)) = frame1(p).  Returns img1, img2 and a
    function mapping frame-1 (u,v) to the true frame-2 position."""
    tex = texture(nv, nu, seed, margin=margin)
    vv, uu = np.mgrid[0:nv, 0:nu].astype(np.float64)
    img1 = ndimage.map_coordinates(tex, [vv + margin, uu + margin], order=3, mode="nearest")
    cu, cv = (nu - 1) / 2.0, (nv - 1) / 2.0
    s = 1.0 + zoom
    # q = c + s (p - c) + shift  =>  p = c + (q - c - shift)/s
    pu = cu + (uu - cu - shift[0]) / s
    pv = cv + (vv - cv - shift[1]) / s
    img2 = ndimage.map_coordinates(tex, [pv + margin, pu + margin], order=3, mode="nearest")
    if noise > 0:
        r = _rng(seed + 7)
        img1 = img1 + r.normal(0, noise, img1.shape)
        img2 = img2 + r.normal(0, noise, img2.shape)
    img1 = np.clip(np.rint(img1), 0, 255).astype(np.uint8)
    img2 = np.clip(np.rint(img2), 0, 255).astype(np.uint8)

    def truth(u, v):
        return cu + s * (np.asarray(u) - cu) + shift[0], cv + s * (np.asarray(v) - cv) + shift[1]

    return img1, img2, truth


def frame_sequence(n_frames, nv=376, nu=1241, seed=1023, max_shift=6.0, zoom=0.005, noise=1.0):
    """frame k+1 = frame k warped by U[-6,6]^2 px + 0.5 % zoom + N(0,1) noise.  Each
    frame is resampled from the base texture with the accumulated warp (margin sized for it)."""
    r = _rng(seed + 1)
    margin = int(np.ceil(n_frames * (max_shift + zoom * max(nu, nv)))) + 8
    tex = texture(nv, nu, seed, margin=margin)
    vv, uu = np.mgrid[0:nv, 0:nu].astype(np.float64)
    cu, cv = (nu - 1) / 2.0, (nv - 1) / 2.0
    frames, warps = [], []
    S, tu, tv = 1.0, 0.0, 0.0  # q = c + S (p - c) + t  maps frame-0 coords to frame-k coords
    for k in range(n_frames):
        pu = cu + (uu - cu - tu) / S
        pv = cv + (vv - cv - tv) / S
        img = ndimage.map_coordinates(tex, [pv + margin, pu + margin], order=3, mode="nearest")
        if noise > 0:
            img = img + _rng(seed + 100 + k).normal(0, noise, img.shape)
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        warps.append((S, tu, tv))
        du, dv = r.uniform(-max_shift, max_shift, 2)
        S, tu, tv = S * (1 + zoom), (1 + zoom) * tu + du, (1 + zoom) * tv + dv
    return frames, warps


def keypoints(n, nv=376, nu=1241, seed=2047, border=20.5):
    """(u,v) ~ U[border, n-1-border], float32-representable (41x41 windows interior)."""
    r = _rng(seed)
    u = r.uniform(border, nu - 1 - border, n).astype(np.float32).astype(np.float64)
    v = r.uniform(border, nv - 1 - border, n).astype(np.float32).astype(np.float64)
    return u, v


def rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def pose43(R_cw, C):
    """MATLAB pose [R_WC; t_WC] (modules/VisualOdometry.m:318): orientation = R_cw, location = C."""
    return np.vstack([R_cw, np.asarray(C, dtype=np.float64).reshape(1, 3)])


def project(K, R_cw, C, X):
    xc = (X - C) @ R_cw.T
    uvw = xc @ K.T
    return uvw[:, :2] / uvw[:, 2:3], xc[:, 2]


def pnp_scene(n, seed=4095, K=KITTI_K, nv=376, nu=1241, noise=0.3, outlier_frac=0.2):
    """landmarks back-projected from image points at Z~U[5,100] under the identity
    pose; true pose = 0.8 m forward + 1 deg yaw; observations = projections + N(0,noise) px with a
    fraction replaced by uniform gross outliers.  Image points use the K convention (x,y)."""
    r = _rng(seed)
    x = r.uniform(20, nu - 20, n); y = r.uniform(20, nv - 20, n)
    Z = r.uniform(5, 100, n)
    X = np.stack([(x - K[0, 2]) / K[0, 0] * Z, (y - K[1, 2]) / K[1, 1] * Z, Z], 1)
    R_cw = rot_y(1.0)
    C = np.array([0.0, 0.0, 0.8])
    xy, depth = project(K, R_cw, C, X)
    xy = xy + r.normal(0, noise, xy.shape)
    n_out = int(round(outlier_frac * n))
    out_idx = r.permutation(n)[:n_out]
    xy[out_idx, 0] = r.uniform(0, nu, n_out)
    xy[out_idx, 1] = r.uniform(0, nv, n_out)
    is_out = np.zeros(n, bool); is_out[out_idx] = True
    return xy, X, R_cw, C, is_out


def triangulation_scene(n, seed=8191, K=KITTI_K, nv=376, nu=1241, noise=0.2, n_poses=10, step=0.8):
    """C = n candidates whose first observations come from `n_poses` poses along a
    straight track (temporal recall 10, modules/VisualOdometry.m:23)."""
    r = _rng(seed)
    x = r.uniform(20, nu - 20, n); y = r.uniform(20, nv - 20, n)
    Z = r.uniform(5, 100, n)
    # landmarks defined in the CURRENT camera frame (pose1), then moved to world
    R1 = rot_y(2.0); C1 = np.array([0.3, 0.0, step * n_poses])
    Xc = np.stack([(x - K[0, 2]) / K[0, 0] * Z, (y - K[1, 2]) / K[1, 1] * Z, Z], 1)
    X = Xc @ R1 + C1
    pose1 = pose43(R1, C1)
    which = r.integers(0, n_poses, n)
    poses0 = np.zeros((n, 4, 3)); xy0 = np.zeros((n, 2))
    for k in range(n_poses):
        Rk = rot_y(0.2 * k); Ck = np.array([0.02 * k, 0.0, step * k])
        sel = which == k
        poses0[sel] = pose43(Rk, Ck)
        p, _ = project(K, Rk, Ck, X[sel])
        xy0[sel] = p
    xy1, _ = project(K, R1, C1, X)
    xy0 = xy0 + r.normal(0, noise, xy0.shape)
    xy1 = xy1 + r.normal(0, noise, xy1.shape)
    return xy0, xy1, poses0, pose1, X


def plane_sequence(n_frames, K, nv=240, nu=320, seed=77, depth=12.0, step=(0.12, 0.02, 0.10), yaw_deg=0.15, noise=0.5, sigma=2.0):
    """A 3D-consistent sequence for the frame loop (modules/VisualOdometry.m:264-346): a textured
    fronto-parallel plane Z = depth (world = camera 0) seen by a camera that moves by `step` metres
    and turns by `yaw_deg` per frame.  Returns frames (uint8), poses (4x3 [R_cw; C] each) and a
    function lifting frame-0 pixels to their landmarks on the plane.  sigma: the texture's blur (2 px: ~13k min-eigen
    corners on a KITTI-sized frame; 1 px: four times as many)."""
    margin = 96
    tex = texture(nv, nu, seed, sigma=sigma, margin=margin)
    vv, uu = np.mgrid[0:nv, 0:nu].astype(np.float64)
    Kinv = np.linalg.inv(K)
    rays = np.stack([uu.ravel(), vv.ravel(), np.ones(uu.size)], 0)
    r = _rng(seed + 1)
    frames, poses = [], []
    for k in range(n_frames):
        R_cw = rot_y(yaw_deg * k)
        C = np.array(step, dtype=np.float64) * k
        d = R_cw.T @ (Kinv @ rays)                      # ray directions in world coordinates
        lam = (depth - C[2]) / d[2]
        X = C[:, None] + lam * d                         # intersection with the plane
        tu = K[0, 0] * X[0] / depth + K[0, 2]            # the plane's texture = frame 0
        tv = K[1, 1] * X[1] / depth + K[1, 2]
        img = ndimage.map_coordinates(tex, [tv.reshape(nv, nu) + margin, tu.reshape(nv, nu) + margin], order=3, mode="nearest")
        if noise > 0:
            img = img + r.normal(0, noise, img.shape)
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        poses.append(pose43(R_cw, C))

    def lift(u, v):
        return np.stack([(u - K[0, 2]) / K[0, 0] * depth, (v - K[1, 2]) / K[1, 1] * depth, np.full_like(u, depth, dtype=np.float64)], 1)

    return frames, poses, lift


def relief_sequence(n_frames, K, nv=240, nu=320, seed=91, depth=12.0, relief=2.5, step=(0.25, 0.03, 0.05), yaw_deg=0.2,
                    noise=0.5):
    """A non-planar 3D-consistent sequence (the bootstrap's eight-point solver is degenerate on a
    plane): a textured height field Z(u0, v0) = depth + relief * smooth noise over the image plane of
    view 0 (frame 0 = the texture), seen by a camera that moves by `step` and turns by `yaw_deg` per
    frame.  Returns frames, poses (4x3 [R_cw; C]) and lift(u0, v0) -> landmarks."""
    margin = 96
    tex = texture(nv, nu, seed, margin=margin)
    r = _rng(seed + 7)
    zmap = ndimage.gaussian_filter(r.standard_normal(tex.shape), 28.0)
    zmap = depth + relief * zmap / np.abs(zmap).max()
    vv, uu = np.mgrid[0:nv, 0:nu].astype(np.float64)
    rays = np.linalg.inv(K) @ np.stack([uu.ravel(), vv.ravel(), np.ones(uu.size)], 0)
    frames, poses = [], []
    for k in range(n_frames):
        R_cw = rot_y(yaw_deg * k)
        C = np.array(step, dtype=np.float64) * k
        d = R_cw.T @ rays
        lam = (depth - C[2]) / d[2]
        for _ in range(12):                                   # ray / height-field intersection, fixed point
            X = C[:, None] + lam * d
            u0 = K[0, 0] * X[0] / X[2] + K[0, 2]; v0 = K[1, 1] * X[1] / X[2] + K[1, 2]
            zs = ndimage.map_coordinates(zmap, [v0 + margin, u0 + margin], order=1, mode="nearest")
            lam = (zs - C[2]) / d[2]
        X = C[:, None] + lam * d
        u0 = K[0, 0] * X[0] / X[2] + K[0, 2]; v0 = K[1, 1] * X[1] / X[2] + K[1, 2]
        img = ndimage.map_coordinates(tex, [v0.reshape(nv, nu) + margin, u0.reshape(nv, nu) + margin], order=3, mode="nearest")
        if noise > 0:
            img = img + r.normal(0, noise, img.shape)
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        poses.append(pose43(R_cw, C))

    def lift(u, v):
        z = ndimage.map_coordinates(zmap, [np.asarray(v) + margin, np.asarray(u) + margin], order=1, mode="nearest")
        return np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)

    return frames, poses, lift


def stress_state(K, lift, new_state, n_l=25000, n_c=25000, nv=376, nu=1241, seed=5, step=(0.2, 0.0, 0.15), yaw=0.15, max_age=10):
    """The frame loop's state at the stress size (SURVEY 8d stress variant; bench.py `frame_loop.at_50k_keypoints`):
    n_l landmarks on the plane of plane_sequence (lift: frame-0 pixels -> landmarks), and n_c candidates first seen
    1 .. max_age frames before frame 0 -- poses extrapolated backwards along the track, first keypoints = their
    projections there.  new_state: the constructor of the state record (modules._new_state)."""
    rng = np.random.default_rng(seed)
    lu = rng.uniform(25, nu - 26, n_l); lv = rng.uniform(25, nv - 26, n_l)
    st = new_state()
    st.keypoints = np.stack([lu, lv], 1); st.landmarks = lift(lu, lv); st.reproError = np.zeros(n_l)
    cu0 = rng.uniform(25, nu - 26, n_c); cv0 = rng.uniform(25, nv - 26, n_c)
    Xc = lift(cu0, cv0)
    age = rng.integers(1, max_age + 1, n_c)
    step = np.asarray(step, dtype=np.float64)
    first_xy = np.zeros((n_c, 2)); first_pose = np.zeros((n_c, 4, 3))
    for a_ in range(1, max_age + 1):
        sel = age == a_
        R_, C_ = rot_y(-yaw * a_), -step * a_
        first_pose[sel] = pose43(R_, C_)
        first_xy[sel] = project(K, R_, C_, Xc[sel])[0]
    st.candidate_keypoints = np.stack([cu0, cv0], 1); st.candidate_first_keypoints = first_xy
    st.candidate_first_poses = first_pose; st.candidate_time_indxs = -age.astype(np.int64)
    return st
