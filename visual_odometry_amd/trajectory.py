"""Trajectory output and evaluation.

The reference keeps the poses of a run in a 4 x 3 x N array (module_tests/VisualOdometry_test.m:72-74,101) and
draws the top view from [pose(4,1) pose(4,3)] per frame (utils/VOVisualizer.m:111-112) beside the loaders'
ground truth, which is the same two numbers of the true pose ([end-8 end] of a 12-column row,
utils/loadGeneralData.m:13-14).  It computes no error figure; monocular VO fixes scale and frame only up to a
similarity, so the evaluation here aligns the top view to the ground truth with the closed-form least-squares
similarity (Umeyama 1991) before taking the RMSE.  Host arithmetic on a few numbers per frame: no kernel.
"""
import numpy as np


class Trajectory:
    """poses[:, :, i] = [R_WC; t_WC] of frame i, as in VisualOdometry_test.m:73-74,101."""

    def __init__(self, first_pose=None):
        p0 = np.vstack([np.eye(3), np.zeros((1, 3))]) if first_pose is None else np.asarray(first_pose, np.float64)
        self._poses = [p0.reshape(4, 3).copy()]
        self.status = [0]

    def append(self, pose, status=0):
        self._poses.append(np.asarray(pose, np.float64).reshape(4, 3).copy())
        self.status.append(int(status))

    def __len__(self):
        return len(self._poses)

    @property
    def poses(self):
        """4 x 3 x N, MATLAB's layout of the array."""
        return np.stack(self._poses, -1)

    @property
    def topViewTrajectory(self):
        """N x 2 [x z] of the camera centres (utils/VOVisualizer.m:111-112)."""
        return np.array([[p[3, 0], p[3, 2]] for p in self._poses])

    @property
    def centres(self):
        return np.array([p[3] for p in self._poses])

    def kitti_rows(self):
        """N x 12: each pose as a row-major 3 x 4 [R_WC' t'] row, the format of the datasets' pose files, so that
        `rows[:, [end-8 end]]` is the top view again (utils/loadGeneralData.m:13-14)."""
        return np.array([np.hstack([p[:3].T, p[3][:, None]]).reshape(-1) for p in self._poses])

    def save(self, path):
        np.savetxt(path, self.kitti_rows(), fmt="%.9e")

    @staticmethod
    def load(path):
        rows = np.loadtxt(path, ndmin=2)
        t = Trajectory(_pose_of_row(rows[0]))
        for r in rows[1:]:
            t.append(_pose_of_row(r))
        return t


def _pose_of_row(r):
    m = np.asarray(r, np.float64).reshape(3, 4)
    return np.vstack([m[:, :3].T, m[:, 3]])


def umeyama(src, dst, with_scale=True):
    """Least-squares similarity dst ~ s R src + t (Umeyama 1991), any dimension.  -> (s, R, t)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    if src.shape != dst.shape or src.ndim != 2 or len(src) < 2:
        raise ValueError("umeyama: two N x d point sets of equal shape, N >= 2")
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    cov = b.T @ a / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(src.shape[1])
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[-1, -1] = -1.0
    R = U @ S @ Vt
    var = (a ** 2).sum() / len(src)
    s = float((D * np.diag(S)).sum() / var) if with_scale and var > 0 else 1.0
    return s, R, mu_d - s * R @ mu_s


def evaluate(estimated_xz, ground_truth_xz):
    """Top view against ground truth after similarity alignment.  -> dict(rmse, max, scale, aligned, length)."""
    e, g = np.asarray(estimated_xz, np.float64), np.asarray(ground_truth_xz, np.float64)
    n = min(len(e), len(g))
    e, g = e[:n], g[:n]
    s, R, t = umeyama(e, g)
    al = (s * (R @ e.T)).T + t
    d = np.linalg.norm(al - g, axis=1)
    return {"rmse": float(np.sqrt((d ** 2).mean())), "max": float(d.max()), "scale": s, "aligned": al,
            "length": float(np.linalg.norm(np.diff(g, axis=0), axis=1).sum())}
