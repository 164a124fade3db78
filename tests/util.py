import numpy as np


def rot_angle(Ra, Rb):
    """Geodesic angle between two rotation matrices (rad)."""
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def pose_diff(Ta, Tb):
    Ta = np.asarray(Ta).reshape(4, 4)
    Tb = np.asarray(Tb).reshape(4, 4)
    return float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])), rot_angle(Ta[:3, :3], Tb[:3, :3])


# north_star tolerance: poses match the reference CPU path to <= 1e-4 m / <= 1e-3 rad
POS_TOL_M = 1e-4
ROT_TOL_RAD = 1e-3
