"""Host-side SE(3) helpers the hot path's wrappers need.

Only the semantics of the reference's ``trans_mat`` (numbotics/math/spatial.py:157-178),
``rot_diff`` (:207-212, the definition that wins at import time) and URDF rpy handling
(``Rotation.from_euler('xyz', rpy)``, physics/helpers.py:312-316) are restated; this is setup
math, not a kernel.
"""
import numpy as np


def rpy_matrix(rpy) -> np.ndarray:
    """Extrinsic x-y-z (URDF fixed-axis roll/pitch/yaw) rotation: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr = np.cos(r), np.sin(r)
    cp, sp = np.cos(p), np.sin(p)
    cy, sy = np.cos(y), np.sin(y)
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cr, -sr], [0.0, sr, cr]])
    Ry = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]])
    Rz = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
    return Rz @ Ry @ Rx


def trans_mat(pos=np.zeros((3,)), orn=np.eye(3)) -> np.ndarray:
    """Homogeneous transform(s) from position(s) and rotation matrix/matrices."""
    pos = np.asarray(pos, dtype=np.float64)
    orn = np.asarray(orn, dtype=np.float64)
    if pos.shape[-1] != 3:
        raise ValueError(f"Position must have 3 elements, got {pos.shape[-1]}")
    if orn.shape[-2:] != (3, 3):
        raise ValueError(f"Orientation must be a 3x3 matrix, got {orn.shape}")
    if pos.ndim != orn.ndim - 1:
        raise ValueError("Position and orientation must have the same number of batch dimensions")
    T = np.zeros(orn.shape[:-2] + (4, 4))
    T[..., :3, :3] = orn
    T[..., :3, 3] = pos
    T[..., 3, 3] = 1.0
    return T


def skew_to_vec(S: np.ndarray) -> np.ndarray:
    return np.stack([S[..., 2, 1], S[..., 0, 2], S[..., 1, 0]], axis=-1)


def rot_diff(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """vee(0.5 (R - R^T)) with R = B A^T (reference: math/spatial.py:207-212)."""
    R = B @ np.swapaxes(A, -2, -1)
    return skew_to_vec(0.5 * (R - np.swapaxes(R, -2, -1)))
