"""Trajectory bookkeeping and evaluation: host logic, no GPU."""
import numpy as np
import pytest

from visual_odometry_amd.trajectory import Trajectory, evaluate, umeyama


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def test_pose_array_top_view_and_file_round_trip(tmp_path):
    t = Trajectory()
    for k in range(1, 6):
        t.append(np.vstack([rot_y(0.01 * k), [0.1 * k, -0.02 * k, 0.8 * k]]), status=k % 2)
    assert t.poses.shape == (4, 3, 6) and np.array_equal(t.poses[:, :, 0], np.vstack([np.eye(3), np.zeros((1, 3))]))
    np.testing.assert_allclose(t.topViewTrajectory[3], [0.3, 2.4])          # [pose(4,1) pose(4,3)], VOVisualizer.m:111-112
    rows = t.kitti_rows()
    assert rows.shape == (6, 12)
    np.testing.assert_allclose(rows[:, [12 - 9, 12 - 1]], t.topViewTrajectory)   # [end-8 end], loadGeneralData.m:13-14
    t.save(tmp_path / "poses.txt")
    back = Trajectory.load(tmp_path / "poses.txt")
    np.testing.assert_allclose(back.poses, t.poses, atol=1e-9)
    assert t.status == [0, 1, 0, 1, 0, 1]


def test_similarity_alignment_recovers_a_known_transform():
    rng = np.random.default_rng(4)
    src = rng.normal(size=(40, 2))
    a = 0.7
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    dst = 3.5 * src @ R.T + np.array([2.0, -1.0])
    s, Rr, t = umeyama(src, dst)
    np.testing.assert_allclose(s, 3.5, rtol=1e-12); np.testing.assert_allclose(Rr, R, atol=1e-12)
    np.testing.assert_allclose(t, [2.0, -1.0], atol=1e-12)
    ev = evaluate(src, dst)
    assert ev["rmse"] < 1e-12 and abs(ev["scale"] - 3.5) < 1e-12
    noisy = dst + rng.normal(scale=0.05, size=dst.shape)
    assert 0.02 < evaluate(src, noisy)["rmse"] < 0.1
    s3, R3, t3 = umeyama(np.c_[src, src[:, 0] ** 2], np.c_[src, src[:, 0] ** 2] @ rot_y(0.3).T * 2 + 1)
    np.testing.assert_allclose(s3, 2.0, rtol=1e-12); np.testing.assert_allclose(R3, rot_y(0.3), atol=1e-12)
    refl = src * np.array([1.0, -1.0])                                       # a mirror image is not a rotation
    assert np.linalg.det(umeyama(src, refl)[1]) > 0
    with pytest.raises(ValueError):
        umeyama(src[:1], dst[:1])
