"""Procedural scenes (host side, numpy).

The reference ships no assets (`scenes*`, `*.obj`, `*.xml` are git-ignored), so the
benchmark and test workloads are generated here, following SURVEY §8(d):
  hello_world  (cfg1)  cube on a ground plane, one spot light, 256x256
  vocalfold    (cfg2-4) larynx tube 64x128 quads + two 96x96-quad vocal-fold lips = 53,248 tris,
                        50 animation frames of one oscillation cycle (same topology every frame,
                        like the OBJ sequences of main.py:84-85)
  colon        (cfg5)  curved tube with haustral ripples, 256x1024 quads = 524,288 tris
Conventions are Mitsuba's [EXT, SURVEY App. A]: a sensor looks down +z of its local frame,
`perspective_projection` maps camera space to [0,1]^2 sample space.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


# ----------------------------------------------------------------------------- conventions
def perspective_projection(width, height, fov_x_deg, near, far):
    """mi.perspective_projection(film_size, crop_size=film_size, crop_offset=0, fov_x, near, far)
    [EXT, SURVEY App. A]; call site examples/vocalfold_scene.py:31-38."""
    aspect = width / height
    c = 1.0 / np.tan(np.deg2rad(fov_x_deg) * 0.5)
    return np.array(
        [
            [-0.5 * c, 0.0, 0.5, 0.0],
            [0.0, -0.5 * aspect * c, 0.5, 0.0],
            [0.0, 0.0, far / (far - near), -near * far / (far - near)],
            [0.0, 0.0, 1.0, 0.0],
        ],
        dtype=np.float32,
    )


def look_at(origin, target, up=(0.0, 1.0, 0.0)):
    """Mitsuba's Transform4f.look_at: columns = (left, new_up, dir, origin)."""
    o = np.asarray(origin, np.float64)
    d = np.asarray(target, np.float64) - o
    d /= np.linalg.norm(d)
    left = np.cross(np.asarray(up, np.float64), d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m.astype(np.float32)


# ----------------------------------------------------------------------------- containers
@dataclass
class MeshData:
    name: str
    frames: np.ndarray  # [T,V,3] float32 (T = 1 for a static mesh)
    tris: np.ndarray  # [F,3] int32, mesh-local
    albedo: tuple = (0.8, 0.8, 0.8)
    material: str = "mat-Default"
    bsdf: dict = None  # None: diffuse (Lambert); a dict: Mitsuba `principled` parameters by name (missing ones take PRINCIPLED_DEFAULTS)
    # the mesh carries vertex normals (an OBJ with `vn`, a PLY with nx/ny/nz): Mitsuba then shades it in the frame of the
    # interpolated normal and re-derives angle-weighted vertex normals after every vertex_positions update (include/ffx.h ffx_smooth)
    smooth: bool = False
    # per-vertex texture coordinates [V,2] (an OBJ's `vt`, v already flipped as Mitsuba's flip_tex_coords default does) and the
    # material's texture-valued base colour [h,w,3] (Mitsuba: <bsdf> with a bitmap `base_color`; parameter key
    # `<mat>.brdf_0.base_color.data`, re-assigned per iteration by the reference's dataset loop, main.py:120-153)
    uv: np.ndarray = None
    base_tex: np.ndarray = None


@dataclass
class SensorData:
    name: str
    to_world: np.ndarray
    fov_x: float
    near: float
    far: float
    width: int
    height: int

    @property
    def K(self):
        return perspective_projection(self.width, self.height, self.fov_x, self.near, self.far)


@dataclass
class SpotData:
    name: str
    to_world: np.ndarray
    intensity: tuple = (10.0, 10.0, 10.0)
    cutoff_angle: float = 20.0
    beam_width: float = 15.0


@dataclass
class SceneData:
    meshes: List[MeshData]
    camera: SensorData
    projector: Optional[SensorData] = None  # film size = texture size
    spot: Optional[SpotData] = None
    projector_scale: float = 1.0
    notes: dict = field(default_factory=dict)

    @property
    def n_tris(self):
        return sum(m.tris.shape[0] for m in self.meshes)


# ----------------------------------------------------------------------------- primitives
def grid_tris(nu, nv, wrap_u=False):
    """Triangles of an (nu x nv)-quad grid whose vertices are laid out [iv*(nu_v) + iu]."""
    nuv = nu if wrap_u else nu + 1
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = iv * nuv + iu
    b = iv * nuv + (iu + 1) % nuv
    c = (iv + 1) * nuv + iu
    d = (iv + 1) * nuv + (iu + 1) % nuv
    t = np.stack([np.stack([a, b, d], -1), np.stack([a, d, c], -1)], -2)
    return t.reshape(-1, 3).astype(np.int32)


def make_plane(z, half, nu=1, nv=1):
    u = np.linspace(-half, half, nu + 1)
    v = np.linspace(-half, half, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    verts = np.stack([uu, vv, np.full_like(uu, z)], -1).reshape(-1, 3).astype(np.float32)
    return verts, grid_tris(nu, nv)


def make_cube(center, half):
    c = np.asarray(center, np.float32)
    s = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * half + c
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in q:
        tris += [(a, b, cc), (a, cc, d)]
    return s, np.asarray(tris, np.int32)


def make_uv_sphere(center, radius, nu=32, nv=16):
    th = np.linspace(0, np.pi, nv + 1)
    ph = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    pp, tt = np.meshgrid(ph, th, indexing="xy")
    verts = np.stack([np.sin(tt) * np.cos(pp), np.sin(tt) * np.sin(pp), np.cos(tt)], -1).reshape(-1, 3)
    verts = (verts * radius + np.asarray(center)).astype(np.float32)
    return verts, grid_tris(nu, nv, wrap_u=True)


# ----------------------------------------------------------------------------- cfg1
def hello_world(width=256, height=256):
    gv, gt = make_plane(0.0, 4.0, 1, 1)
    # ground plane in the xz-plane at y = 0
    gv = gv[:, [0, 2, 1]].copy()
    cv, ct = make_cube((0.0, 0.5, 0.0), 0.5)
    cam = SensorData("PerspectiveCamera", look_at((2.5, 2.0, 3.0), (0.0, 0.4, 0.0)), 45.0, 0.01, 100.0, width, height)
    spot = SpotData("emit-Spot", look_at((1.0, 4.0, 2.0), (0.0, 0.0, 0.0), up=(0, 0, 1)), (40.0, 40.0, 40.0), 35.0, 25.0)
    return SceneData(
        [MeshData("mesh-Ground", gv[None], gt, (0.6, 0.6, 0.6)), MeshData("mesh-Cube", cv[None], ct, (0.8, 0.3, 0.2))],
        cam, None, spot, notes={"config": "cfg1 hello_world"},
    )


# ----------------------------------------------------------------------------- cfg2-4
def _tube(n_around, n_along, z0, z1, radius_fn):
    th = np.linspace(0, 2 * np.pi, n_around, endpoint=False)
    zz = np.linspace(z0, z1, n_along + 1)
    tt, zg = np.meshgrid(th, zz, indexing="xy")
    r = radius_fn(tt, zg)
    verts = np.stack([r * np.cos(tt), r * np.sin(tt), zg], -1).reshape(-1, 3).astype(np.float32)
    return verts, grid_tris(n_around, n_along, wrap_u=True)


def _fold_lip(side, phase, n=96, R=1.25, z0=5.0, half_len=1.15, g0=0.28):
    """One vocal-fold lip as an (n x n)-quad sheet: u runs anterior->posterior (y), v runs from the
    lateral wall over the superior surface to the medial edge and down the medial surface."""
    u = np.linspace(-1.0, 1.0, n + 1)
    v = np.linspace(0.0, 1.0, n + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    y = uu * half_len
    taper = 1.0 - uu**2  # glottis closes at the commissures
    vs = 0.6  # v < vs: superior surface; v >= vs: medial surface
    s_sup = np.clip(vv / vs, 0, 1)
    s_med = np.clip((vv - vs) / (1 - vs), 0, 1)
    # mucosal wave: the lower margin leads the upper margin
    open_up = max(0.0, np.sin(phase))
    open_lo = max(0.0, np.sin(phase + 0.9))
    gap_up = 0.02 + g0 * open_up * taper
    gap_lo = 0.02 + g0 * open_lo * taper
    gap = gap_up * (1 - s_med) + gap_lo * s_med + 0.25 * s_med**2
    wall = np.sqrt(np.maximum(R**2 - y**2, 0.05))
    x_sup = wall * (1 - s_sup) + gap_up * s_sup
    x = np.where(vv < vs, x_sup, gap)
    bulge = 0.12 * np.sin(np.pi * s_sup) * (0.6 + 0.4 * open_up)
    z = np.where(vv < vs, z0 + 0.35 * (1 - s_sup) - bulge, z0 + 1.1 * s_med + 0.0)
    verts = np.stack([side * x, y, z], -1).reshape(-1, 3).astype(np.float32)
    tris = grid_tris(n, n)
    if side < 0:
        tris = tris[:, [0, 2, 1]]
    return verts, tris


def vocalfold(width=512, height=512, tex=500, frames=50, n_fold=96, tube=(64, 128), principled=True):
    """cfg2-4: F = 2*tube[0]*tube[1] + 2 * 2*n_fold^2 = 16384 + 36864 = 53248 triangles.
    The material is a principled BSDF with the plugin's defaults, like the reference's "mat-Default OBJ" whose
    `brdf_0.specular` examples/vocalfold_scene.py:93 randomises (principled=False: diffuse)."""
    bsdf = {} if principled else None
    lar_v, lar_t = _tube(tube[0], tube[1], 0.8, 7.5, lambda t, z: 1.25 + 0.08 * np.sin(3 * t) * np.sin(1.3 * z) + 0.05 * np.cos(2.1 * z))
    fold_frames = []
    for k in range(frames):
        ph = 2 * np.pi * k / frames
        lv, lt = _fold_lip(+1, ph, n_fold)
        rv, rt = _fold_lip(-1, ph, n_fold)
        fold_frames.append(np.concatenate([lv, rv], 0))
    fold_t = np.concatenate([lt, rt + lv.shape[0]], 0).astype(np.int32)
    cam = SensorData("PerspectiveCamera", look_at((0.0, 0.0, 1.5), (0.0, 0.0, 5.0)), 60.0, 0.01, 100.0, width, height)
    proj = SensorData("PerspectiveCamera_1", look_at((0.25, 0.0, 1.5), (0.0, 0.0, 5.0)), 30.0, 0.01, 100.0, tex, tex)
    spot = SpotData("emit-Spot", look_at((0.0, 0.1, 1.5), (0.0, 0.0, 5.0)), (8.0, 8.0, 8.0), 40.0, 30.0)
    return SceneData(
        [
            MeshData("mesh-Larynx", lar_v[None], lar_t, (0.80, 0.32, 0.34), "mat-Default OBJ", bsdf),
            MeshData("mesh-VocalFold", np.stack(fold_frames).astype(np.float32), fold_t, (0.85, 0.62, 0.60), "mat-Default OBJ", bsdf),
        ],
        cam, proj, spot, projector_scale=20.0, notes={"config": "vocalfold", "frames": frames},
    )


# ----------------------------------------------------------------------------- cfg5
def colon(width=1024, height=1024, tex=1024, n_around=256, n_along=1024, principled=True):
    """cfg5: curved tube with haustral ripples, 2*256*1024 = 524,288 triangles; material "mat-Mucosa", a principled BSDF
    as in the reference's main.py:97-107 (principled=False: diffuse)."""
    th = np.linspace(0, 2 * np.pi, n_around, endpoint=False)
    s = np.linspace(0.0, 1.0, n_along + 1)
    tt, ss = np.meshgrid(th, s, indexing="xy")
    bend = 0.9 * ss  # radians of the torus segment
    Rc = 9.0
    r = 1.4 + 0.22 * np.sin(2 * np.pi * 14 * ss) ** 2 + 0.06 * np.sin(3 * tt + 9 * ss)
    cx = Rc * (1 - np.cos(bend))
    cz = Rc * np.sin(bend)
    nx, nz = np.cos(bend), -np.sin(bend)  # in-plane normal of the centre line
    x = cx + r * np.cos(tt) * nx
    z = cz + r * np.cos(tt) * nz
    y = r * np.sin(tt)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    tris = grid_tris(n_around, n_along, wrap_u=True)
    cam = SensorData("PerspectiveCamera", look_at((0.0, 0.0, 0.3), (0.35, 0.0, 3.0)), 90.0, 0.01, 100.0, width, height)
    proj = SensorData("PerspectiveCamera_1", look_at((0.2, 0.0, 0.3), (0.35, 0.0, 3.0)), 60.0, 0.01, 100.0, tex, tex)
    spot = SpotData("emit-Spot", look_at((0.0, 0.1, 0.3), (0.35, 0.0, 3.0)), (6.0, 6.0, 6.0), 60.0, 45.0)
    return SceneData([MeshData("mesh-Colon", verts[None], tris, (0.85, 0.45, 0.40), "mat-Mucosa", {} if principled else None)], cam, proj, spot,
                     projector_scale=6.0, notes={"config": "colon"})


# ----------------------------------------------------------------------------- materials
# Mitsuba 3.5 `principled` plugin defaults [EXT: plugin documentation]; `specular` is converted to `eta` the way the plugin does
PRINCIPLED_DEFAULTS = {"roughness": 0.5, "anisotropic": 0.0, "metallic": 0.0, "spec_trans": 0.0, "specular": 0.5, "spec_tint": 0.0, "sheen": 0.0,
                       "sheen_tint": 0.0, "flatness": 0.0, "clearcoat": 0.0, "clearcoat_gloss": 0.0}
# column of each parameter in a material row (include/ffx.h FFX_MAT_*)
MAT_STRIDE = 16
MAT_COLUMN = {"model": 3, "roughness": 4, "anisotropic": 5, "metallic": 6, "spec_trans": 7, "eta": 8, "spec_tint": 9, "sheen": 10, "sheen_tint": 11,
              "flatness": 12, "clearcoat": 13, "clearcoat_gloss": 14}


def specular_to_eta(specular):
    """eta = 2 / (1 - sqrt(0.08 specular)) - 1  [EXT: principled.cpp constructor / parameters_changed]"""
    return 2.0 / (1.0 - float(np.sqrt(0.08 * float(specular)))) - 1.0


def material_row(albedo, bsdf):
    """one row of the material table (include/ffx.h): [base_color(3), model, roughness, ..., clearcoat_gloss, 0]"""
    row = np.zeros(MAT_STRIDE, np.float32)
    row[0:3] = albedo
    if bsdf is None:
        row[MAT_COLUMN["eta"]] = 1.0
        return row
    p = dict(PRINCIPLED_DEFAULTS)
    p.update({k: v for k, v in bsdf.items() if k != "eta"})
    row[MAT_COLUMN["model"]] = 1.0
    for k, col in MAT_COLUMN.items():
        if k in p:
            row[col] = float(p[k])
    row[MAT_COLUMN["eta"]] = float(bsdf["eta"]) if "eta" in bsdf and "specular" not in bsdf else specular_to_eta(p["specular"])
    return row


def material_rows(scene):
    """[S,16] material rows, or None if every mesh is diffuse and untextured (the renderer then takes the [S,3] albedo table).
    Column FFX_MAT_BASE_TEX = 1 + index of the mesh's material among base_textures(scene)."""
    if all(m.bsdf is None and m.base_tex is None for m in scene.meshes):
        return None
    rows = np.stack([material_row(m.albedo, m.bsdf) for m in scene.meshes]).astype(np.float32)
    tex_mats = [k for k, _ in base_textures(scene)]
    for i, m in enumerate(scene.meshes):
        if m.base_tex is not None and m.uv is not None and m.material in tex_mats:
            rows[i, 15] = 1.0 + tex_mats.index(m.material)
    return rows


def base_textures(scene):
    """[(material name, [h,w,3] float32)] — one texture-valued base colour per material (first mesh that declares one; at
    most 4: include/ffx.h FFX_MAX_BASE_TEX)"""
    out, seen = [], set()
    for m in scene.meshes:
        if m.base_tex is not None and m.uv is not None and m.material not in seen:
            seen.add(m.material)
            out.append((m.material, np.ascontiguousarray(m.base_tex, np.float32).reshape(m.base_tex.shape[0], m.base_tex.shape[1], 3)))
    if len(out) > 4:
        raise NotImplementedError("more than 4 textured base colours in one scene")
    return out


def slot_uv_table(order, tris, tri_shape, meshes):
    """[F + 4, 6] float32: the texture coordinates (u0 v0 u1 v1 u2 v2) of every LEAF SLOT's triangle (`order`: slot -> triangle,
    ffx_bvh_info.off_order); zeros for meshes without uv"""
    order, tris, tri_shape = np.asarray(order, np.int64), np.asarray(tris, np.int64), np.asarray(tri_shape, np.int64)
    out = np.zeros((order.shape[0] + 4, 6), np.float32)
    for s_, m in enumerate(meshes):
        if m.uv is None:
            continue
        uv = np.asarray(m.uv, np.float32).reshape(-1, 2)
        if uv.shape[0] != m.frames.shape[1]:
            raise ValueError(f"{m.name}: {uv.shape[0]} texture coordinates for {m.frames.shape[1]} vertices")
        slots = np.nonzero(tri_shape[order] == s_)[0]
        out[slots] = uv[tris[order[slots]]].reshape(-1, 6)
    return out


# ----------------------------------------------------------------------------- flattening
def flatten(scene: SceneData):
    """-> (pool [P,3], tris [F,3] mesh-local, tri_shape [F], frame0_off [S], frame_stride [S], n_frames [S], albedo [S,3])."""
    pools, tris, shape, off, stride, nfr, alb = [], [], [], [], [], [], []
    base = 0
    for sidx, m in enumerate(scene.meshes):
        T, V = m.frames.shape[0], m.frames.shape[1]
        pools.append(m.frames.reshape(-1, 3))
        tris.append(m.tris)
        shape.append(np.full(m.tris.shape[0], sidx, np.int32))
        off.append(base)
        stride.append(V)
        nfr.append(T)
        alb.append(m.albedo)
        base += T * V
    return (
        np.concatenate(pools, 0).astype(np.float32), np.concatenate(tris, 0).astype(np.int32), np.concatenate(shape, 0),
        np.asarray(off, np.int32), np.asarray(stride, np.int32), np.asarray(nfr, np.int32), np.asarray(alb, np.float32),
    )


def smooth_tables(tris, tri_shape, smooth, n_shapes):
    """Host tables of include/ffx.h `ffx_smooth` for shape-local triangles `tris` [F,3] with shapes `tri_shape` [F] and one flag
    per shape: -> (shape_smooth [S] int32, shape_vbase [S] int32, adj_start [n_vn + 1] int32, adj [*] int32, n_vn).
    Vertex row = vbase[shape] + local index; adj lists, per row, the incident corners `triangle << 2 | corner` in ascending
    order (flat shapes get rows but no entries)."""
    tris, tri_shape = np.asarray(tris, np.int64), np.asarray(tri_shape, np.int64)
    flags = np.asarray([1 if f else 0 for f in smooth], np.int32)
    if flags.shape[0] != n_shapes:
        raise ValueError("one smooth flag per shape")
    if tris.shape[0] >= 1 << 29:
        raise ValueError("too many triangles for the corner keys")
    n_local = np.zeros(n_shapes, np.int64)
    np.maximum.at(n_local, tri_shape, tris.max(axis=1) + 1)
    vbase = np.concatenate([[0], np.cumsum(n_local)[:-1]]).astype(np.int64)
    n_vn = int(n_local.sum())
    sel = np.nonzero(flags[tri_shape] != 0)[0]
    rows = (vbase[tri_shape[sel]][:, None] + tris[sel]).reshape(-1)
    keys = ((sel[:, None] << 2) | np.arange(3)[None, :]).reshape(-1)
    order = np.lexsort((keys, rows))
    adj = keys[order].astype(np.int32)
    adj_start = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_vn))]).astype(np.int32)
    return flags, vbase.astype(np.int32), adj_start, adj, n_vn
