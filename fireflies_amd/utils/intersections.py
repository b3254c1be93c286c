"""fireflies/utils/intersections.py — batched ray/plane and sphere/sphere helpers."""
import torch


def rayPlane(laserOrigin, laserDirection, planeOrigin, planeNormal):
    """distance t [N,1] along each ray to the plane (intersections.py:5-11); nearly parallel rays get the
    reference's `denom / denom`: 1 for a tiny denominator, NaN for an exactly zero one (pinned by golden g12)."""
    denom = torch.sum(planeNormal * laserDirection, dim=1)
    denom = torch.where(torch.abs(denom) < 0.000001, denom / denom, denom)
    t = torch.sum((planeOrigin - laserOrigin) * planeNormal, dim=1) / denom
    return t[:, None]


def sphereSphere(a_coords, a_radius, b_coords, b_radius):
    """[N,1] bool: do the spheres overlap (intersections.py:26-33)."""
    squared_dist = (a_coords - b_coords).pow(2).sum(dim=1, keepdim=True)
    return squared_dist <= (a_radius + b_radius).pow(2)
