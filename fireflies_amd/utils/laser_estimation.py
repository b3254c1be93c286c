"""Pattern initialisers — fireflies/utils/laser_estimation.py.

`initialize_laser(mitsuba_scene, mitsuba_params, firefly_scene, config, mode, device)` returns a
Laser whose pattern is RANDOM, POISSON (blue noise), GRID, or SMARTY (Poisson-disk sampling whose
radius shrinks where the depth varies most over randomised scenes).  The reference's module does not
run at HEAD (undefined names `transforms`, `firefly_scene.projector`, cv2 / Mitsuba only; SURVEY
App. B); the functions here keep its names and intended semantics.  The depth maps come from the HIP
BVH (K7); everything else is small host/torch code (one-off initialisation, off the hot path).
"""
import math

import numpy as np
import torch

from .. import mi
from ..graphics import depth as ffdepth
from ..projection import Laser
from ..sampling import poisson
from . import intersections
from . import math as ffmath


def probability_distribution_from_depth_maps(depth_maps, uniform_weight: float = 0.0):
    """per-pixel standard deviation of the depth over the maps, plus a uniform floor (:25-32)."""
    return depth_maps.std(dim=0) + uniform_weight if isinstance(depth_maps, torch.Tensor) else depth_maps.std(axis=0) + uniform_weight


def points_from_probability_distribution(prob_distribution, num_samples: int):
    """flat pixel indices drawn without replacement with probability ~ map (:35-42)."""
    return prob_distribution.flatten().multinomial(num_samples, replacement=False)


def _sensor_rays(sensor, pos01):
    """sensor.sample_ray for sample positions pos01 [N,2] in [0,1]^2 -> (origins on the near plane,
    unit directions), world space (the Mitsuba calls at :45-145)."""
    scene = sensor._scene
    w, h = scene._film_size[sensor._key]
    K = mi.perspective_projection((w, h), (w, h), (0, 0), sensor.x_fov(), sensor.near_clip(), sensor.far_clip()).matrix.torch()[0].double()
    to_world = sensor.world_transform().matrix.torch()[0].double()
    p = torch.cat([pos01.double().cpu(), torch.zeros(pos01.shape[0], 1, dtype=torch.float64), torch.ones(pos01.shape[0], 1, dtype=torch.float64)], dim=1)
    q = p @ torch.linalg.inv(K).T
    near_p = q[:, :3] / q[:, 3:]
    d_l = near_p / near_p.norm(dim=1, keepdim=True)
    d = d_l @ to_world[:3, :3].T
    o = to_world[:3, 3][None] + d * (sensor.near_clip() / d_l[:, 2:3])
    return o.float(), d.float()


def get_camera_direction(sensor, device=None):
    o, d = _sensor_rays(sensor, torch.tensor([[0.5, 0.5]]))
    return o.to(device) if device else o, d.to(device) if device else d


def get_camera_frustum(sensor, device=None):
    o, d = _sensor_rays(sensor, torch.tensor([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]))
    return (o.to(device), d.to(device)) if device else (o, d)


def getRayFromSensor(sensor, ray_coordinate_in_ndc):
    return _sensor_rays(sensor, torch.tensor([[float(ray_coordinate_in_ndc[0]), float(ray_coordinate_in_ndc[1])]]))


def create_rays(sensor, points):
    """rays through the pixel corners of flat pixel indices `points` (:113-145)."""
    w, h = sensor.film().size()
    idx = points.reshape(-1).cpu().long()
    pos = torch.stack([(idx % w).float() / w, (idx // w).float() / h], dim=1)
    return _sensor_rays(sensor, pos)


def laser_from_ndc_points(sensor, laser_origin, depth_maps, chosen_points, device=torch.device("cuda")):
    """unit directions from the laser origin to the points where camera rays through the chosen
    pixels meet the plane at the mean scene depth (:148-174)."""
    ray_o, ray_d = create_rays(sensor, chosen_points)
    cam_o, cam_d = get_camera_direction(sensor)
    cam_o = sensor.world_transform().matrix.torch()[0][:3, 3][None]
    cam_d = cam_d / cam_d.norm(dim=-1, keepdim=True)
    mean_depth = float(depth_maps.mean())
    plane_origin = cam_o + cam_d * mean_depth
    t = intersections.rayPlane(ray_o, ray_d, plane_origin, -cam_d)
    world_points = ray_o + ray_d * t
    laser_dir = world_points - laser_origin.reshape(1, 3).cpu()
    return (laser_dir / laser_dir.norm(dim=-1, keepdim=True)).to(device)


def generate_epipolar_constraints(scene, params, device):
    """mask [H,W] of the camera pixels inside the convex hull of the projector frustum's far-plane
    corners (and origin) projected into the camera (:187-250; scipy hull, polygon fill by half-plane
    tests instead of cv2.fillPoly)."""
    from scipy.spatial import ConvexHull

    cam, proj = scene.sensors()[0], scene.sensors()[1]
    o, d = get_camera_frustum(proj)
    pts = torch.cat([o + proj.far_clip() * d, o[:1]], dim=0)
    w, h = cam.film().crop_size()
    K = mi.perspective_projection((w, h), (w, h), (0, 0), cam.x_fov(), cam.near_clip(), cam.far_clip()).matrix.torch()[0]
    cam_world = cam.world_transform().matrix.torch()[0]
    local = ffmath.transform_points(pts, torch.linalg.inv(cam_world))
    keep = local[:, 2] > 1e-6
    uv = ffmath.transform_points(local[keep], K)[:, 0:2].numpy() * np.array([w, h])
    if uv.shape[0] < 3:
        return torch.zeros((h, w), dtype=torch.uint8, device=device)
    hull = ConvexHull(uv)
    poly = uv[hull.vertices]  # counter-clockwise
    yy, xx = np.mgrid[0:h, 0:w]
    inside = np.ones((h, w), bool)
    for a, b in zip(poly, np.roll(poly, -1, axis=0)):
        inside &= ((b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])) >= 0
    return torch.from_numpy(inside.astype(np.uint8)).to(device)


def initialize_laser(mitsuba_scene, mitsuba_params, firefly_scene, config, mode, device):
    """(:253-391).  config: n_beams, and for SMARTY n_depthmaps, variational_epsilon,
    smarty_min_radius, smarty_max_radius."""
    proj = mitsuba_scene.sensors()[1]
    near_clip, far_clip = proj.near_clip(), proj.far_clip()
    laser_fov = float(mitsuba_params[proj.id() + ".x_fov"])
    size = proj.film().size()
    LASER_K = mi.perspective_projection(size, proj.film().crop_size(), proj.film().crop_offset(), laser_fov, near_clip, far_clip).matrix.torch()[0]
    n_beams = config.n_beams
    projector = getattr(firefly_scene, "_projector", None) or getattr(firefly_scene, "projector")
    if mode == "RANDOM":
        local = Laser.generate_random_rays(num_beams=n_beams, intrinsic_matrix=LASER_K, device=device)
    elif mode == "POISSON":
        local = Laser.generate_blue_noise_rays(image_size_x=int(size[0]), image_size_y=int(size[1]), num_beams=n_beams, intrinsic_matrix=LASER_K, device=device)
    elif mode == "GRID":
        g = int(math.sqrt(n_beams))
        local = Laser.generate_uniform_rays_by_count(num_beams_x=g, num_beams_y=g, intrinsic_matrix=LASER_K, device=device)
    elif mode == "SMARTY":
        depth_maps = ffdepth.random_depth_maps(firefly_scene, mitsuba_scene, num_maps=config.n_depthmaps)
        variance_map = ffmath.normalize(probability_distribution_from_depth_maps(depth_maps, config.variational_epsilon))
        sampling = variance_map / variance_map.sum()
        radius = config.smarty_min_radius + (config.smarty_max_radius - config.smarty_min_radius) * (1 - ffmath.normalize(sampling))
        # the map is [H,W]; bridson walks it as [x,y], so hand it the transpose
        _, pts = poisson.bridson(radius.T.detach().cpu().numpy().astype(np.float64), 50, rng=getattr(config, "rng", None))
        pts = torch.tensor(np.asarray(pts)).floor().long()
        w = sampling.shape[1]
        # keep samples inside the epipolar mask of the projector frustum (the reference computes this
        # mask and then leaves the multiplication commented out, :330-350; without it most samples of a
        # wide camera image map to beams outside a narrow projector)
        constraint = generate_epipolar_constraints(mitsuba_scene, mitsuba_params, "cpu")
        pts = pts[constraint[pts[:, 1], pts[:, 0]] > 0]
        chosen = pts[:, 1] * w + pts[:, 0]  # flat index = y * W + x
        laser_world = projector.world().cpu()
        laser_dir = laser_from_ndc_points(mitsuba_scene.sensors()[0], laser_world[0:3, 3], depth_maps, chosen, device="cpu")
        # world direction -> projector-local, then into the Laser's storage convention (-FLIP_Y: the
        # physical local direction of a stored ray r is (-r.x, r.y, -r.z), see projection/laser.py)
        loc = ffmath.transform_directions(laser_dir, torch.linalg.inv(laser_world))
        local = (loc * torch.tensor([-1.0, 1.0, -1.0])).to(device)
        local = local / local.norm(dim=-1, keepdim=True)
        # ... and only beams that really are inside the projector frustum
        KF = LASER_K.double() @ torch.diag(torch.tensor([1.0, -1.0, 1.0, 1.0], dtype=torch.float64))
        q = torch.cat([local.double().cpu(), torch.ones(local.shape[0], 1, dtype=torch.float64)], dim=1) @ KF.T
        uv = q[:, :2] / q[:, 3:]
        ok = ((uv > 0.02) & (uv < 0.98)).all(dim=1)
        local = local[ok.to(local.device)]
    else:
        raise ValueError(f"unknown initialisation mode {mode!r}")
    return Laser(projector, local, LASER_K, laser_fov, near_clip, far_clip, device=device)
