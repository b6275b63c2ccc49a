"""Rotation matrices and the geodesic pose error (reference nemo/utils/pose.py:6-59,
nemo/utils/evaluate.py:8-36).  Host side, float64: the validator of the 0.5 degree tolerance."""
import math

import numpy as np


def get_transformation_matrix(azimuth, elevation, distance):
    if distance == 0:
        distance = 0.1
    cam = np.array([[distance * math.cos(elevation) * math.sin(azimuth)],
                    [-distance * math.cos(elevation) * math.cos(azimuth)],
                    [distance * math.sin(elevation)]])
    a, e = -azimuth, -(math.pi / 2 - elevation)
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, math.cos(e), -math.sin(e)], [0, math.sin(e), math.cos(e)]])
    rot = rx @ rz
    out = np.eye(4)
    out[:3, :3] = rot
    out[:3, 3:] = -rot @ cam
    return out


def rotation_theta(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def cal_rotation_matrix(theta, elev, azum, dis):
    if dis <= 1e-10:
        dis = 0.5
    return rotation_theta(theta) @ get_transformation_matrix(azum, elev, dis)[0:3, 0:3]


def pose_error(gt, pred):
    """Geodesic distance (radians) between the rotations of two {azimuth, elevation, theta} dicts."""
    from scipy.linalg import logm
    if pred is None:
        return np.pi
    mats = [cal_rotation_matrix(float(d["theta"]), float(d["elevation"]), float(d["azimuth"]), 5.0) for d in (gt, pred)]
    if any(np.any(np.isnan(m)) or np.any(np.isinf(m)) for m in mats):
        return np.pi
    return ((logm(mats[1].T @ mats[0]) ** 2).sum()) ** 0.5 / (2.0 ** 0.5)
