"""Small matrix helpers with the reference's names and conventions (utils/matrix.py:4-45)."""
import numpy as np


def skew(v: np.ndarray) -> np.ndarray:
    """3-vector -> 3x3 cross-product matrix, float32 like the reference."""
    x, y, z = (v[i] for i in range(3))
    out = np.zeros((3, 3), dtype=np.float32)
    out[0, 1], out[0, 2] = -z, y
    out[1, 0], out[1, 2] = z, -x
    out[2, 0], out[2, 1] = -y, x
    return out


def skew_3(a: float, b: float, c: float) -> np.ndarray:
    return skew(np.array([a, b, c]))


def unskew(m: np.ndarray) -> np.ndarray:
    return np.array([m[2, 1], m[0, 2], m[1, 0]], dtype=np.float32).reshape(3, 1)


def hom_inv(T: np.ndarray) -> np.ndarray:
    """Inverse of a rigid 4x4 transform (R^T, -R^T t), dtype preserved."""
    assert T.shape == (4, 4), "T has to be a homogeneous transform matrix "
    out = np.zeros_like(T)
    Rt = T[:3, :3].T
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ T[:3, 3]
    out[3, 3] = 1
    return out


def to_hom(pts: np.ndarray) -> np.ndarray:
    return np.concatenate([pts, np.ones((pts.shape[0], 1), dtype=pts.dtype)], axis=1)

