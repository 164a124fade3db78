import numpy as np


def rot_angle(Ra, Rb):
    """Geodesic angle between two rotation matrices (rad)."""
    # ||Ra - Rb||_F = 2*sqrt(2)*sin(angle/2): accurate for tiny angles (arccos of the trace is not)
    return float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ra - Rb) / (2.0 * np.sqrt(2.0)))))


def pose_diff(Ta, Tb):
    Ta = np.asarray(Ta).reshape(4, 4)
    Tb = np.asarray(Tb).reshape(4, 4)
    return float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])), rot_angle(Ta[:3, :3], Tb[:3, :3])


# north_star tolerance: poses match the reference CPU path to <= 1e-4 m / <= 1e-3 rad
POS_TOL_M = 1e-4
ROT_TOL_RAD = 1e-3
