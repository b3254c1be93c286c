"""Laser — fireflies/projection/laser.py: the optimisable point pattern.

`_rays [N,3]` are unit directions in the projector's local frame, stored with z < 0 and
reconciled with the +z-forward sensor convention through FLIP_Y (laser.py:31-33,262-275).
projectRaysToNDC() (K1) and generateTexture() (K2) are differentiable HIP calls; the no-grad
constraint projections after an optimiser step (clamp_to_fov, randomize_*_out_of_bounds,
normalize_rays; laser.py:199-255) reuse the same kernels.

Methods that raise at the reference's HEAD because of a half-finished refactor (SURVEY F8:
rays/origin/originPerRay/setToWorld dereference `self._fireflies...`; generate_uniform_rays_by_count,
generate_random_rays, randomize_laser_out_of_bounds, render_epipolar_lines use the empty
utils.transforms) are implemented with their evident semantics.
"""
import math
from typing import List

import numpy as np
import torch
import yaml

from . import camera
from .. import functional as Fn
from .. import ops
from ..utils import math as ffmath

_FLIP_Y = np.diag([1.0, -1.0, 1.0, 1.0]).astype(np.float32)


def _unit(t):
    return t / torch.linalg.norm(t, dim=-1, keepdims=True)


class Laser(camera.Camera):
    # ------------------------------------------------------------------ pattern generators
    @staticmethod
    def generate_uniform_rays(intra_ray_angle: float, num_beams_x: int, num_beams_y: int, device=torch.device("cuda")):
        """grid of directions (tan((x-(nx-1)/2) a), tan((y-(ny-1)/2) a), -1), normalised
        (laser.py:19-37).  Tangents are evaluated in double and cast, like the reference.  Rows are
        ordered x-major; the reference's index x*nx + y is only right for square grids and is
        replaced by x*ny + y (identical when nx == ny)."""
        rows = np.zeros((num_beams_x * num_beams_y, 3), np.float32)
        for x in range(num_beams_x):
            for y in range(num_beams_y):
                rows[x * num_beams_y + y] = (math.tan((x - (num_beams_x - 1) / 2) * intra_ray_angle),
                                             math.tan((y - (num_beams_y - 1) / 2) * intra_ray_angle), -1.0)
        rays = torch.from_numpy(rows).to(device)
        return rays / torch.linalg.norm(rays, dim=-1, keepdims=True)

    @staticmethod
    def _unproject(points_screen, intrinsic_matrix):
        inv = torch.linalg.inv(intrinsic_matrix.detach().to("cpu", torch.float64)).to(torch.float32)
        rays = _unit(ffmath.transform_points(points_screen.to("cpu"), inv))
        rays[:, 2] *= -1.0
        return rays

    @staticmethod
    def generate_uniform_rays_by_count(num_beams_x: int, num_beams_y: int, intrinsic_matrix, device=torch.device("cuda")):
        """regular grid of screen points at the bin centres, un-projected (laser.py:39-66)."""
        xs = torch.arange((1 / num_beams_x) / 2, 1, 1 / num_beams_x)
        ys = torch.arange((1 / num_beams_y) / 2, 1, 1 / num_beams_y)
        xy = torch.stack(torch.meshgrid(xs, ys, indexing="ij")).movedim(0, -1).reshape(-1, 2)
        pts = torch.cat([xy, -torch.ones(xy.shape[0], 1)], dim=1)
        return Laser._unproject(pts, intrinsic_matrix).to(device)

    @staticmethod
    def generate_random_rays(num_beams: int, intrinsic_matrix, device=torch.device("cuda")):
        """random screen points within +-0.05 of the centre, un-projected (laser.py:68-92)."""
        pts = torch.ones([num_beams, 3], device=device) * 0.5 + (torch.rand([num_beams, 3], device=device) - 0.5) / 10.0
        pts[:, 2] = -1.0
        return Laser._unproject(pts, intrinsic_matrix).to(device)

    @staticmethod
    def generate_blue_noise_rays(image_size_x: int, image_size_y: int, num_beams: int, intrinsic_matrix, device=torch.device("cuda")):
        """Poisson-disk screen points (radius chosen for about num_beams samples, inflated by 25 %),
        un-projected (laser.py:94-145).  The count is approximate, as in the reference."""
        from ..sampling import poisson

        radius = math.sqrt((image_size_x * image_size_y) / (math.pi * num_beams))
        radius += radius / 4.0
        _, samples = poisson.bridson(np.ones([image_size_x, image_size_y]) * radius)
        pts = torch.tensor(np.asarray(samples), dtype=torch.float32) / torch.tensor([image_size_x, image_size_y], dtype=torch.float32)
        pts = torch.cat([pts, -torch.ones(pts.shape[0], 1)], dim=1)
        return Laser._unproject(pts, intrinsic_matrix).to(device)

    # ------------------------------------------------------------------ construction
    def __init__(self, transformable, ray_directions, perspective, max_fov: float, near_clip: float = 0.01, far_clip: float = 1000.0,
                 device=torch.device("cuda")):
        super().__init__(transformable, perspective, max_fov, near_clip, far_clip, device)
        self._rays = ray_directions.to(self.device)
        self.device = device
        K = perspective.detach().to("cpu", torch.float32).numpy().reshape(4, 4)
        self._KF = (K @ _FLIP_Y).astype(np.float32)  # host constant of K1
        self._KF_inv = np.linalg.inv(self._KF.astype(np.float64)).astype(np.float32)

    def rays(self):
        """world-space ray directions (laser.py:163-167)."""
        return ffmath.transform_directions(self._rays, self._transformable.world().to(self._rays.device))

    def origin(self):
        return self._transformable.world()

    def originPerRay(self):
        return self._transformable.world()[0:3, 3].unsqueeze(0).repeat(self._rays.shape[0], 1)

    def setToWorld(self, to_world) -> None:
        self._transformable.set_world(to_world)

    def near_clip(self) -> float:
        return self._near_clip

    def far_clip(self) -> float:
        return self._far_clip

    # ------------------------------------------------------------------ K1
    def projectRaysToNDC(self):
        """[N,3] screen-space points, xy in [0,1]^2 inside the frustum (laser.py:262-275).
        Differentiable w.r.t. `_rays`."""
        return Fn.project_rays(self._rays, self._KF)

    def projectNDCPointsToWorld(self, points):
        """inverse of projectRaysToNDC (laser.py:277-290); no gradient."""
        return ops.transform_points(points.detach().contiguous(), self._KF_inv, 0)

    # ------------------------------------------------------------------ constraint projection (no grad)
    def normalize(self, tensor):
        return _unit(tensor)

    @torch.no_grad()
    def normalize_rays(self) -> None:
        self._rays[:] = _unit(self._rays)

    @torch.no_grad()
    def initRandomRays(self):
        pts = torch.rand(self._rays.shape, device=self.device) * 2.0 - 1.0
        pts[:, 2] = 1.0
        self._rays = _unit(self.projectNDCPointsToWorld(pts))

    def initPoissonDiskSamples(self, width, height, radius):
        return None

    @torch.no_grad()
    def clamp_to_fov(self, clamp_val: float = 0.95, epsilon: float = 0.0001, then_normalize: bool = False) -> None:
        """project, clamp the screen position to [1 - clamp_val, clamp_val], un-project, normalise
        (laser.py:199-206) — one launch, in place.  then_normalize=True also applies the
        normalize_rays() that the training loops call right after (laser.py:254-255)."""
        if self._rays.is_cuda and self._rays.is_contiguous():
            ops.clamp_to_fov_(self._rays.detach(), self._KF, self._KF_inv, 1 - clamp_val, clamp_val, 2 if then_normalize else 1)
            self._edits = getattr(self, "_edits", 0) + 1  # (a native in-place edit: torch's version counter does not see it — optim.PatternOptimizer._premade_key)
            return
        ndc = ops.project_rays_fwd(self._rays.detach().contiguous(), self._KF)
        ndc[:, 0:2] = torch.clamp(ndc[:, 0:2], 1 - clamp_val, clamp_val)
        self._rays[:] = _unit(self.projectNDCPointsToWorld(ndc))
        if then_normalize:
            self.normalize_rays()

    @torch.no_grad()
    def _respawn(self, out_of_bounds) -> None:
        n = int(out_of_bounds.sum())
        if n == 0:
            return
        pts = torch.rand((n, 3), device=self.device)
        pts[:, 2] = -1.0
        new_rays = self._rays.clone()
        new_rays[out_of_bounds] = self.projectNDCPointsToWorld(pts)
        self._rays[:] = _unit(new_rays)

    @torch.no_grad()
    def randomize_laser_out_of_bounds(self) -> None:
        """respawn points that left the projector frustum (laser.py:208-231)."""
        ndc = ops.transform_points(self._rays.detach().contiguous(), self._perspective.detach().to("cpu").numpy(), 0)
        xy = ndc[:, 0:2]
        self._respawn(((xy >= 1.0) | (xy <= 0.0)).any(dim=1))

    @torch.no_grad()
    def randomize_camera_out_of_bounds(self, ndc_coords) -> None:
        """respawn points whose camera-space NDC left (-1,1) (laser.py:233-249)."""
        xy = ndc_coords[:, 0:2]
        self._respawn(((xy >= 1.0) | (xy <= -1.0)).any(dim=1))

    # ------------------------------------------------------------------ K2
    def generateTexture(self, sigma: float, texture_size: List[int], reduce: str = None, half_window: int = -1):
        """rasterize_points(projectRaysToNDC()[:, :2], sigma, texture_size) (laser.py:292-296).
        The reference forces this onto the CPU and returns the dense [N,H,W] stack; here it stays on
        the device.  reduce="sum"/"softor" returns the fused [H,W] texture instead (what every
        caller computes next, vocalfold_scene.py:59) without materialising the stack."""
        size = [int(v) for v in (texture_size.tolist() if isinstance(texture_size, torch.Tensor) else texture_size)]
        pts = self.projectRaysToNDC()[:, 0:2].contiguous()
        if reduce is None:
            return Fn.rasterize_points_dense(pts, sigma, size[0], size[1])
        return Fn.splat(pts, sigma, size[0], size[1], reduce, half_window)

    def render_epipolar_lines(self, sigma: float, texture_size, camera_world=None):
        """soft epipolar segments of every beam in the CAMERA image (laser.py:298-325): the segment
        between the near- and far-clip points of the beam, projected with this laser's K."""
        from ..graphics import rasterization

        o, d = self.originPerRay(), self.rays()
        cam = self._transformable.world() if camera_world is None else camera_world
        w2c = torch.linalg.inv(cam.to("cpu")).to(o.device)
        K = self._perspective.to(o.device)
        ends = []
        for t in (self._near_clip, self._far_clip):
            p = ffmath.transform_points(ffmath.transform_points(o + t * d, w2c), K)[:, 0:2]
            ends.append(p)
        return rasterization.rasterize_lines(torch.stack(ends, dim=1), sigma, texture_size, device=o.device)

    # ------------------------------------------------------------------ persistence
    def save(self, filepath: str):
        """YAML {rays, fov, near_clip, far_clip} (laser.py:327-336)."""
        with open(filepath, "w") as f:
            yaml.dump({"rays": self._rays.detach().cpu().numpy().tolist(), "fov": float(self._fov), "near_clip": float(self._near_clip),
                       "far_clip": float(self._far_clip)}, f)

    @staticmethod
    def load_rays(filepath: str, device=torch.device("cuda")):
        """counterpart of save() (the reference has no loader)."""
        with open(filepath, "r") as f:
            d = yaml.safe_load(f)
        return torch.tensor(d["rays"], dtype=torch.float32, device=device), d
