"""4x4 transform helpers — same names, argument meaning and results as the reference's
fireflies/utils/math.py (file:line cited per function).  Small host/device glue in torch; the
per-vertex hot path does NOT go through here (it is fused into ffx_scene_update)."""
import math
import random

import torch
import torch.nn.functional as F


def uniformBetweenValues(a: float, b: float) -> float:  # utils/math.py:8-9
    return random.uniform(a, b)


def _rot(rows, device):
    return torch.tensor(rows, device=device)


def getYawTransform(alpha: float, _device) -> torch.Tensor:
    """rotation about Z (utils/math.py:24-34)."""
    c, s = math.cos(alpha), math.sin(alpha)
    return _rot([[c, -s, 0], [s, c, 0], [0, 0, 1]], _device)


def getPitchTransform(alpha: float, _device) -> torch.Tensor:
    """rotation about Y (utils/math.py:37-47)."""
    c, s = math.cos(alpha), math.sin(alpha)
    return _rot([[c, 0, s], [0, 1, 0], [-s, 0, c]], _device)


def getRollTransform(alpha: float, _device) -> torch.Tensor:
    """rotation about X (utils/math.py:50-60)."""
    c, s = math.cos(alpha), math.sin(alpha)
    return _rot([[1, 0, 0], [0, c, -s], [0, s, c]], _device)


# aliases, utils/math.py:12-21
def getZTransform(alpha, _device):
    return getYawTransform(alpha, _device)


def getYTransform(alpha, _device):
    return getPitchTransform(alpha, _device)


def getXTransform(alpha, _device):
    return getRollTransform(alpha, _device)


def vector_dot(A, B):  # utils/math.py:63-64
    return (A * B).sum(dim=-1)


def rotation_matrix_from_vectors(v1, v2):
    """Rodrigues rotation taking v1 onto v2 (utils/math.py:67-105)."""
    a = F.normalize(v1, dim=0)
    b = F.normalize(v2, dim=0)
    axis = torch.linalg.cross(a, b)
    cos = torch.dot(a, b)
    zero = torch.zeros((), device=a.device, dtype=torch.float32)
    kx = torch.stack(
        [torch.stack([zero, -axis[2], axis[1]]), torch.stack([axis[2], zero, -axis[0]]), torch.stack([-axis[1], axis[0], zero])]
    ).to(torch.float32)
    return torch.eye(3, device=a.device) + kx + (kx @ kx) * (1 - cos) / torch.norm(axis) ** 2


def singleRandomBetweenTensors(a, b):
    """one python-RNG draw shared by all components (utils/math.py:162-167).  The reference adds
    `b` instead of `a` there; this returns the evident intent a + r (b - a)."""
    assert a.size() == b.size() and a.device == b.device
    return random.uniform(0, 1) * (b - a) + a


def randomBetweenTensors(a, b):  # utils/math.py:170-175
    assert a.size() == b.size() and a.device == b.device
    return torch.rand(a.shape, device=a.device) * (b - a) + a


def normalize(tensor):  # utils/math.py:178-181 (min-max normalisation)
    tensor = tensor - tensor.amin()
    return tensor / tensor.amax()


def normalize_channelwise(tensor, dim: int = -1, device=None):  # utils/math.py:184-196
    dims = [d for d in range(tensor.dim()) if d != (dim % tensor.dim())]
    tensor = tensor - tensor.amin(dims)
    return tensor / tensor.amax(dims)


def convert_points_to_homogeneous(points):  # utils/math.py:199-200
    return F.pad(points, pad=(0, 1), mode="constant", value=1.0)


def convert_points_from_homogeneous(points):  # utils/math.py:212-213
    return points[..., :-1] / points[..., -1:]


def convert_points_to_nonhomogeneous(points):  # utils/math.py:216-217
    return F.pad(points, pad=(0, 1), mode="constant", value=0.0)


def toMat4x4(mat, addOne: bool = True):  # utils/math.py:203-209
    out = F.pad(mat, pad=(0, 1, 0, 1), mode="constant", value=0.0)
    if addOne:
        out[3, 3] = 1.0
    return out


def transform_points(points, transform):
    """homogeneous transform with perspective divide (utils/math.py:220-228)."""
    ph = convert_points_to_homogeneous(points)
    q = torch.matmul(transform.unsqueeze(0), ph.unsqueeze(-1)).squeeze(dim=-1)
    return convert_points_from_homogeneous(q)


def transform_directions(points, transform):
    """direction transform, w = 0 (utils/math.py:231-235)."""
    ph = convert_points_to_nonhomogeneous(points)
    q = (transform.unsqueeze(0) @ ph.unsqueeze(-1)).squeeze(-1)
    return q[..., :-1]
