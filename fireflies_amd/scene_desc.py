"""Builds the C-ABI parameter blocks (include/ffx.h: ffx_camera, ffx_scene_desc) from host data."""
import ctypes as C

import numpy as np

from . import _abi


def _m16(m):
    a = np.asarray(m, dtype=np.float32).reshape(-1)
    if a.size != 16:
        raise ValueError("expected a 4x4 matrix")
    return (C.c_float * 16)(*a.tolist())


def camera_struct(to_world, camera_to_sample, near, far, width, height):
    c = _abi.Camera()
    c.to_world = _m16(to_world)
    c.camera_to_sample = _m16(camera_to_sample)
    c.near_clip, c.far_clip, c.width, c.height = float(near), float(far), int(width), int(height)
    return c


def camera_from_sensor(s, to_world=None):
    return camera_struct(s.to_world if to_world is None else to_world, s.K, s.near, s.far, s.width, s.height)


def set_rfilter(sd, rfilter):
    """the film's reconstruction filter: None / "box", "gaussian" or ("gaussian", stddev) — Mitsuba's `gaussian` is hdrfilm's default
    (include/ffx.h, ffx_scene_desc.rfilter); rendered by ffx_render_{fwd,bwd}_filtered"""
    kind, stddev = (rfilter, 0.0) if rfilter is None or isinstance(rfilter, str) else (rfilter[0], float(rfilter[1]))
    if kind in (None, "box"):
        sd.rfilter, sd.rfilter_stddev = _abi.RFILTER_BOX, 0.0
    elif kind == "gaussian":
        if stddev < 0.0 or stddev > 0.5:
            raise ValueError(f"gaussian reconstruction filter: stddev {stddev} outside (0, 0.5] (radius 4 stddev must fit the 5x5-pixel window)")
        sd.rfilter, sd.rfilter_stddev = _abi.RFILTER_GAUSSIAN, stddev
    else:
        raise ValueError(f"reconstruction filter {kind!r}: box and gaussian are implemented")
    return sd


def scene_desc(scene, n_shapes=None, tex_channels=1, color=(0.0, 1.0, 0.0), shadows=True, cam_to_world=None, proj_to_world=None,
               spot_to_world=None, spot_intensity=None, mat_stride=0, base_tex=None, slot_uv=None, host_mats=None, rfilter=None):
    """ffx_scene_desc for a scenes.SceneData.  `color` is the RGB weight of a 1-channel projector
    texture (the reference packs the laser texture into the green channel,
    examples/vocalfold_scene.py:64-67)."""
    sd = _abi.SceneDesc()
    sd.cam = camera_from_sensor(scene.camera, cam_to_world)
    sd.shadows = int(shadows) if isinstance(shadows, int) and not isinstance(shadows, bool) else int(bool(shadows))  # (an int: include/ffx.h FFX_SHADOWS_* bits)
    set_rfilter(sd, rfilter)
    sd.n_shapes = int(n_shapes if n_shapes is not None else len(scene.meshes))
    sd.mat_stride = int(mat_stride)  # 0 / 3: the material table is [S,3] Lambert albedos; 16: material rows (scenes.material_rows)
    if host_mats is not None:  # the material table travels with the call (kernel arguments): [n_shapes, 3 | 16] host array
        set_host_materials(sd, host_mats)
    if base_tex:  # texture-valued base colours: [(address, width, height)] + the address of the per-slot texture coordinates
        if len(base_tex) > _abi.MAX_BASE_TEX or slot_uv is None:
            raise ValueError("at most 4 base-colour textures, and they need slot_uv")
        sd.n_base_tex = len(base_tex)
        for k, (addr, w, h) in enumerate(base_tex):
            sd.base_tex[k], sd.base_tex_w[k], sd.base_tex_h[k] = int(addr), int(w), int(h)
        sd.slot_uv = int(slot_uv)
    if scene.projector is not None:
        p = scene.projector
        sd.proj.to_world = _m16(p.to_world if proj_to_world is None else proj_to_world)
        sd.proj.camera_to_sample = _m16(p.K)
        sd.proj.scale = float(scene.projector_scale)
        sd.proj.color = (C.c_float * 3)(*[float(v) for v in color])
        sd.proj.tex_w, sd.proj.tex_h, sd.proj.tex_channels = int(p.width), int(p.height), int(tex_channels)
        sd.proj.enabled = 1
    if scene.spot is not None:
        s = scene.spot
        sd.spot.to_world = _m16(s.to_world if spot_to_world is None else spot_to_world)
        inten = s.intensity if spot_intensity is None else spot_intensity
        sd.spot.intensity = (C.c_float * 3)(*[float(v) for v in inten])
        sd.spot.cutoff_deg, sd.spot.beam_width_deg = float(s.cutoff_angle), float(s.beam_width)
        sd.spot.enabled = 1
    return sd


def set_host_materials(sd, rows):
    """ffx_scene_desc.mat_h: the rows [n_shapes, stride] (host array) become part of the scene description — the render calls then
    take no material pointer and a randomisation enqueues no upload.  False (nothing set) if the table is too large."""
    a = np.ascontiguousarray(rows, np.float32)
    stride = int(sd.mat_stride) or 3
    if a.ndim != 2 or a.shape[1] != stride or a.shape[0] < sd.n_shapes:
        raise ValueError(f"host material table {a.shape} does not match the scene description (n_shapes {sd.n_shapes}, stride {stride})")
    n = int(sd.n_shapes) * stride
    if n > _abi.MAX_MAT_H:
        return False
    C.memmove(sd.mat_h, a.ctypes.data, 4 * n)
    sd.n_mat_h = n
    return True
