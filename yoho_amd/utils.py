"""Host helpers with the reference's names (utils/utils.py, utils/r_eval.py).  Small numpy utilities
only - the GPU path never routes compute through them."""
import os
import numpy as np


def make_non_exists_dir(fn):
    """utils/utils.py:13-15"""
    if not os.path.exists(fn):
        os.makedirs(fn)


def transform_points(pts, transform):
    """utils/utils.py:42-50"""
    h, w = transform.shape
    if h == 3 and w == 3:
        return pts @ transform.T
    if h == 3 and w == 4:
        return pts @ transform[:, :3].T + transform[:, 3:].T
    elif h == 4 and w == 4:
        hp = np.concatenate([pts, np.ones([pts.shape[0], 1])], 1) @ transform.T
        return hp[:, :-1] / hp[:, -1:]
    else:
        raise NotImplementedError


def to_cuda(data):
    """utils/utils.py:108-136 (dict / list of tensors -> device)."""
    if type(data) == list:
        return [[t.cuda() for t in item] if type(item).__name__ == "list" else item.cuda() for item in data]
    elif type(data) == dict:
        return {k: ([t.cuda() for t in v] if type(v).__name__ == "list" else v.cuda()) for k, v in data.items()}
    raise NotImplementedError


def matrix_from_quaternion(quaternion):
    """utils/r_eval.py:94-110"""
    w, x, y, z = quaternion[0], quaternion[1], quaternion[2], quaternion[3]
    mat = np.eye(3)
    mat[0, 0] = 1 - 2 * y * y - 2 * z * z
    mat[0, 1] = 2 * x * y - 2 * z * w
    mat[0, 2] = 2 * x * z + 2 * y * w
    mat[1, 0] = 2 * x * y + 2 * z * w
    mat[1, 1] = 1 - 2 * x * x - 2 * z * z
    mat[1, 2] = 2 * y * z - 2 * x * w
    mat[2, 0] = 2 * x * z - 2 * y * w
    mat[2, 1] = 2 * y * z + 2 * x * w
    mat[2, 2] = 1 - 2 * x * x - 2 * y * y
    return mat


def dataset_feature_name(name):
    """'3dLomatch/..' shares the feature cache of '3dmatch/..' (tests/extractor.py:84-87, tests/matcher.py:24-27)."""
    return f"3d{name[4:]}" if name[0:4] == "3dLo" else name


def quaternion_from_matrix(matrix):
    """Rotation matrix (3x3 or the 3x3 block of a 4x4) -> unit quaternion (w, x, y, z) with w >= 0: the eigenvector of
    the largest eigenvalue of the symmetric 4x4 matrix K built from the rotation (utils/r_eval.py quaternion_from_matrix,
    isprecise=False branch, i.e. transformations.py / Bar-Itzhack)."""
    M = np.asarray(matrix, dtype=np.float64)
    m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
    m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
    m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
    K = np.array([[m00 - m11 - m22, 0.0, 0.0, 0.0],
                  [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
                  [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
                  [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22]])
    K /= 3.0
    w, V = np.linalg.eigh(K)
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        np.negative(q, q)
    return q
