"""CPU oracle for the augmentation chain.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` (and the fixture generator under ``tests/golden``) may import this file.

A numpy restatement of ``src/data/augment.py`` with every random draw turned into an ARGUMENT (the reference draws from the
global ``np.random`` state inside each function), following the reference's own array arithmetic line by line so that dtypes
behave the same (float32 clouds stay float32 until they meet a float64 rotation matrix).  PINNED:
``tests/golden/make_golden_aug.py`` ran the reference's functions in the build container with ``np.random`` wrapped to record
their draws; ``tests/golden/augment.npz`` holds inputs, draws and outputs; ``tests/test_augment_oracle_golden.py`` checks this
file against them.
"""
import numpy as np


def normalize_point_cloud(points):
    """augment.py:4-18."""
    points = points - np.mean(points, axis=0)
    m = np.max(np.sqrt(np.sum(points ** 2, axis=1)))
    return points / m if m > 0 else points


def random_point_dropout(points, drop_mask):
    """augment.py:20-33 with the draw ``np.random.random(N) <= ratio`` given as a mask."""
    points = points.copy()
    idx = np.where(drop_mask)[0]
    if len(idx) > 0:
        points[idx, :] = points[0, :]
    return points


def random_scale_point_cloud(points, scale):
    """augment.py:35-47."""
    return points * scale


def random_shift_point_cloud(points, shifts):
    """augment.py:49-60."""
    return points + shifts


def random_jitter_point_cloud(points, noise, clip=0.05):
    """augment.py:62-74; ``noise`` = sigma * randn(N,3)."""
    return points + np.clip(noise, -clip, clip)


def rot_y(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_z(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rot_x(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def rotate_point_cloud_y(points, angle):
    """augment.py:76-91 (note: multiplies by the matrix itself, not its transpose)."""
    return points @ rot_y(angle)


def rotate_point_cloud_z(points, angle):
    """augment.py:93-108."""
    return points @ rot_z(angle)


def random_rotate_point_cloud(points, angle_z, angle_y, angle_x):
    """augment.py:158-191: R = Rz Ry Rx, points @ R^T."""
    return points @ (rot_z(angle_z) @ rot_y(angle_y) @ rot_x(angle_x)).T


def drop_and_replace_with_noise(points, drop_indices, uniform01, gauss):
    """augment.py:110-156.  ``uniform01`` (k,3) in [0,1) stands for np.random.uniform(min, max) = min + (max-min) u,
    ``gauss`` (k,3) = noise_std * randn."""
    out = points.copy()
    lo, hi = np.min(points, axis=0), np.max(points, axis=0)
    out[drop_indices] = lo + (hi - lo) * uniform01 + gauss
    return out


def scale_point_cloud(points, scale3, scale_xyz=(True, True, True)):
    """augment.py:193-226 on the 'xyz' entry."""
    s = np.array(scale3, dtype=np.float64).copy()
    for i, on in enumerate(scale_xyz):
        if not on:
            s[i] = 1.0
    return points * s


def center_and_normalize_point_cloud(points, center=True, normalize=True, gravity_dim=1):
    """augment.py:228-259 -> (xyz, heights)."""
    h = points[:, gravity_dim:gravity_dim + 1]
    heights = h - np.min(h, axis=0)
    if center:
        points = points - np.mean(points, axis=0)
    if normalize:
        m = np.max(np.sqrt(np.sum(points ** 2, axis=1)))
        if m > 0:
            points = points / m
    return points, heights


def axis_rotation(axis_idx, theta):
    """expm(cross(eye(3), axis * theta)) of augment.py:292 in closed form: the rotation by theta about a coordinate axis."""
    return (rot_x, rot_y, rot_z)[axis_idx](theta)


def rotate_point_cloud(points, thetas, order):
    """augment.py:261-307: per-axis rotations (theta 0 = identity) multiplied in the drawn order, points @ R^T."""
    R = np.eye(3)
    for i in order:
        R = R @ (axis_rotation(i, thetas[i]) if thetas[i] != 0 else np.eye(3))
    return points @ R.T


def train_chain(points, normalize, rot, scale, noise, clip, drop):
    """scanobjectnn.py:186-205: normalise, then each custom augmentation if its coin came up (None = skipped):
    rot = (az, ay, ax), scale = float, noise = sigma*randn (N,3), drop = (indices, uniform01, gauss)."""
    if normalize:
        points = normalize_point_cloud(points)
    if rot is not None:
        points = random_rotate_point_cloud(points, *rot)
    if scale is not None:
        points = random_scale_point_cloud(points, scale)
    if noise is not None:
        points = random_jitter_point_cloud(points, noise, clip)
    if drop is not None:
        points = drop_and_replace_with_noise(points, *drop)
    return points
