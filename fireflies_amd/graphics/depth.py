"""Primary-visibility queries — fireflies/graphics/depth.py over the HIP BVH (K7).

`scene` is a fireflies_amd.mi.Scene.  Sample indexing, pixel positions (corner, no half-pixel
offset), miss value 0 and the label rule follow depth.py:54-84,119-125; the jitter of
`from_camera` is the counter-based hash of DESIGN.md §4.2 instead of Mitsuba's PCG32 stream.
"""
import torch

from ..utils import math as ffmath


def _trace(scene, spp, jitter, seed=0, want_ids=False):
    """one K7 launch per (pose, camera, sampling): the dataset path asks for the depth map and the segmentation of the
    same randomised scene (main.py:161-166) — the second query reuses the first one's trace (it always carries the ids)"""
    cam = scene.camera_struct(0)
    key = (scene.geom.version, bytes(cam), int(spp), int(jitter), int(seed))
    last = getattr(scene, "_last_trace", None)
    if last is not None and last[0] == key:
        return last[1]
    out = scene.geom.trace_primary(cam, spp, jitter, seed, want_ids=True)
    scene._last_trace = (key, out)
    return out


def from_camera_non_wrapped(scene, spp=64):
    """depth t per (pixel, sample), flat [W*H*spp]; every sample of a pixel is identical because no
    jitter is added (depth.py:49-86)."""
    return _trace(scene, spp, 0)[0]


def from_camera(scene, spp=64, seed=0):
    """same with per-sample jitter (depth.py:128-166)."""
    return _trace(scene, spp, 1, seed)[0]


def _labels(shape_ids):
    # depth.py:119-125 relabels the shape pointers: ids -= ids.min(); ids = ids.max() - ids.  With ptr = id + 1 (0 = no shape, like a null pointer) that is
    # (max ptr - min ptr) - (ptr - min ptr) = max id - id: the minimum cancels — one reduction and one subtraction instead of six launches
    ids = shape_ids.to(torch.int64)
    return ids.max() - ids


def get_segmentation_from_camera(scene, spp=1):
    """shape-label image [H,W] with the reference's relabelling ids -= min; ids = max - ids
    (depth.py:89-125)."""
    _, shape, _ = _trace(scene, spp, 0, want_ids=True)
    w, h = scene.sensors()[0].film().crop_size()
    lab = _labels(shape)
    return lab.reshape(h, w, spp)[..., 0] if spp > 1 else lab.reshape(h, w)


def cast_laser(scene, origin=None, direction=None, laser=None):
    """world-space hit points of laser rays ([N,3]; misses give the origin)."""
    if laser is not None:
        origin, direction = laser.originPerRay(), laser.rays()
    o, d = origin.contiguous().float(), direction.contiguous().float()
    t, _, _ = scene.geom.trace_rays(o, d)
    return o + t.unsqueeze(-1) * d


def cast_laser_id(scene, origin, direction):
    """label of the shape hit by each laser ray, shifted so the smallest is 0 (depth.py:33-46)."""
    _, shape, _ = scene.geom.trace_rays(origin.contiguous().float(), direction.contiguous().float())
    ptr = shape.to(torch.int64) + 1
    return ptr - ptr.min()


def project_to_camera_space(scene, points):
    """world points -> camera NDC in [-1,1] (x,y) with depth; helper that depth.from_laser calls
    but the reference never defined (depth.py:16-17)."""
    cs = scene.camera_struct(0)
    to_world = torch.tensor(list(cs.to_world), dtype=torch.float32).reshape(4, 4)
    K = torch.tensor(list(cs.camera_to_sample), dtype=torch.float32).reshape(4, 4)
    M = (K @ torch.linalg.inv(to_world)).to(points.device)
    s = ffmath.transform_points(points, M)
    return torch.cat([s[:, 0:2] * 2.0 - 1.0, s[:, 2:3]], dim=1).unsqueeze(0)


def from_laser(scene, params, laser):
    """depth map masked to the pixels hit by laser beams (depth.py:9-30)."""
    w, h = scene.sensors()[0].film().size()
    hit = cast_laser(scene, laser=laser)
    ndc = project_to_camera_space(scene, hit)
    pix = torch.floor((ndc[0, :, 0:2] * 0.5 + 0.5) * torch.tensor([w, h], device=hit.device)).long()
    ok = (pix[:, 0] >= 0) & (pix[:, 0] < w) & (pix[:, 1] >= 0) & (pix[:, 1] < h)
    mask = torch.zeros((w, h), device=hit.device)
    mask[pix[ok, 0], pix[ok, 1]] = 1.0
    depth = from_camera_non_wrapped(scene, spp=1).reshape(w, h)
    return depth * mask


def random_depth_maps(firefly_scene, mi_scene, num_maps: int = 100, spp: int = 1):
    """depth maps of `num_maps` randomised scenes, stacked [num_maps,H,W] (depth.py:169-190)."""
    w, h = mi_scene.sensors()[0].film().size()
    out = []
    for _ in range(num_maps):
        firefly_scene.randomize()
        d = from_camera_non_wrapped(mi_scene, spp=spp)
        out.append(d.reshape(h, w, spp).mean(dim=-1))
    return torch.stack(out)
