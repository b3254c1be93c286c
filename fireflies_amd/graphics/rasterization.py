"""Differentiable soft splatting — the public functions of fireflies/graphics/rasterization.py
with the same names, argument order, output shapes and orientation, executed by HIP kernels.

  rasterize_points(points, sigma, texture_size)  -> [N, size1, size0]            (:7-37)
  sum / softor                                                                    (:156-161)
  baked_sum / baked_softor (window-limited), *_2 variants                         (:164-472)
  rasterize_points_baked_sum / _softor (dense, looped over points)                (:475-535)
  rasterize_depth, rasterize_lines, subsampled_point_raster                       (:66-153,538-549)

`texture_size` is (size0, size1) as a tensor or sequence; `device` arguments are accepted for
source compatibility — the tensors must already live on the HIP device.
The fused forms `splat_sum` / `splat_softor` never build the [N,H,W] stack; on the reference this
stack plus ~12 temporaries is the dominant cost of the pattern side.
"""
import math

import torch

from .. import functional as Fn
from .. import ops


def _size(texture_size):
    s = texture_size.tolist() if isinstance(texture_size, torch.Tensor) else list(texture_size)
    return int(s[0]), int(s[1])


def rasterize_points(points, sigma: float, texture_size, device=None):
    s0, s1 = _size(texture_size)
    return Fn.rasterize_points_dense(points[:, 0:2].contiguous() if points.shape[1] != 2 else points, float(sigma), s0, s1)


def rasterize_points_in_non_ndc(points, sigma: float, texture_size, device=None):
    """points already in texel units (:40-62).  Evaluated as rasterize_points(points / size), which
    re-multiplies by the size: the coordinates differ from the reference by one rounding."""
    s0, s1 = _size(texture_size)
    scale = torch.tensor([s0, s1], dtype=points.dtype, device=points.device)
    return rasterize_points(points / scale, sigma, texture_size)


def softor(texture, dim=0, keepdim: bool = False):
    return 1 - torch.prod(1 - texture, dim=dim, keepdim=keepdim)


def sum(texture, dim=0, keepdim: bool = False):  # noqa: A001  (the reference shadows the builtin too)
    return torch.sum(texture, dim=dim, keepdim=keepdim)


def splat_sum(points, sigma: float, texture_size):
    """== sum(rasterize_points(points, sigma, texture_size)) in one kernel."""
    s0, s1 = _size(texture_size)
    return Fn.splat(points, float(sigma), s0, s1, "sum", -1)


def splat_softor(points, sigma: float, texture_size):
    """== softor(rasterize_points(points, sigma, texture_size)) in one kernel."""
    s0, s1 = _size(texture_size)
    return Fn.splat(points, float(sigma), s0, s1, "softor", -1)


def _half_window(sigma, num_std):
    """footprint = floor(sqrt(sigma)) * num_std, made odd; half = (footprint - 1) / 2 (:180-182)."""
    sg = float(sigma.item()) if isinstance(sigma, torch.Tensor) else float(sigma)
    fp = math.floor(math.sqrt(sg)) * num_std
    if fp % 2 == 0:
        fp += 1
    return sg, (fp - 1) // 2


def baked_sum(points, sigma, texture_size, num_std: int = 4, device=None):
    """window-limited sum; `sigma` is the squared width as in the reference's call sites (:577)."""
    s0, s1 = _size(texture_size)
    sg, half = _half_window(sigma, num_std)
    return Fn.splat(points, sg, s0, s1, "sum", half)


def baked_sum_2(points, sigma, texture_size, num_std: int = 4, device=None):
    """the reference's vectorised variant returns the TRANSPOSE of baked_sum (it omits the final
    .T, :318 vs :237); reproduced."""
    return baked_sum(points, sigma, texture_size, num_std).T


def baked_softor(points, sigma, texture_size, num_std: int = 5, device=None):
    s0, s1 = _size(texture_size)
    sg, half = _half_window(sigma, num_std)
    return Fn.splat(points, sg, s0, s1, "softor", half)


def baked_softor_2(points, sigma, texture_size, num_std: int = 5, device=None):
    return baked_softor(points, sigma, texture_size, num_std)


def rasterize_points_baked_softor(points, sigma: float, texture_size, device=None):
    """dense softor accumulated point by point (:475-503) == softor(rasterize_points(...))
    (the reference's version only runs for square textures)."""
    return splat_softor(points, sigma, texture_size)


def rasterize_points_baked_sum(points, sigma: float, texture_size, device=None):
    return splat_sum(points, sigma, texture_size)


def rasterize_depth(points, depth_vals, sigma: float, texture_size, device=None):
    """layers normalised by their own maximum and scaled by depth (:66-104).  Without autograd this
    is one fused kernel; under autograd it is composed from the differentiable dense splat."""
    s0, s1 = _size(texture_size)
    pts = points[:, 0:2].contiguous()
    if torch.is_grad_enabled() and (pts.requires_grad or depth_vals.requires_grad):
        d = Fn.rasterize_points_dense(pts, float(sigma), s0, s1)
        d = d / d.amax(dim=2, keepdim=True).amax(dim=1, keepdim=True)
        return d * depth_vals.reshape(-1, 1).unsqueeze(-1)
    return ops.splat_depth_fwd(pts, depth_vals.reshape(-1).contiguous(), float(sigma), s0, s1)


def rasterize_lines(lines, sigma: float, texture_size, device=None):
    """soft line segments [N,2,2] -> [N,size1,size0] (:107-153), differentiable w.r.t. the segments like the
    reference's torch expression (its line-regularisation loop optimises them, :645-743).  The reference scales
    its INPUT in place (:122-123); this does not mutate the argument."""
    s0, s1 = _size(texture_size)
    return Fn.rasterize_lines(lines, float(sigma), s0, s1)


def subsampled_point_raster(ndc_points, num_subsamples, sigma, sensor_size):
    """multi-resolution pyramid of softor(rasterize_depth) (:538-549)."""
    out = []
    for i in range(num_subsamples):
        r = rasterize_depth(ndc_points[:, 0:2], ndc_points[:, 2:3], sigma, sensor_size // 2**i)
        out.append(softor(r, keepdim=True))
    return out
