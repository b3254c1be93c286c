"""`mi` — the slice of the Mitsuba-3 Python protocol that Fireflies talks to, served by the HIP
path.  Usage: replace `import mitsuba as mi` by `from fireflies_amd import mi`.

What the reference uses: `mi.set_variant`, `mi.traverse(scene)` -> dict-like
parameters with `.update()` (fireflies/scene.py:94-384), value wrappers `mi.Float`, `mi.Float32`,
`mi.Transform4f(...).matrix.torch()`, `mi.TensorXf` (scene.py:135,249,257; examples/
vocalfold_scene.py:69), `scene.sensors()[i].film().size()/crop_size()/crop_offset()`,
`mi.perspective_projection(...)` (vocalfold_scene.py:24-38) and `mi.render(scene, spp=...)`
(:102).  Scene *files* (Mitsuba XML) are a later row: scenes come from
`fireflies_amd.scenes` (procedural) via `mi.load_scene_data`.

Everything numeric lives on the HIP device: `params.update()` runs K5+K6 (ffx_scene_update),
`mi.render` runs K8 (and K9 under autograd when `tex.data` requires grad).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import functional as Fn
from . import _abi, ops, scene_desc, scenes

_variant = "hip_ad_rgb"


_POSE_KEY, _PLAIN_KEY = object(), object()  # Scene._key_plan markers

# float offsets of the fields scene_desc() patches per step (ffx_scene_desc: camera / projector / spot poses, the spot's intensity)
_SD_CAM_TW = (_abi.SceneDesc.cam.offset + _abi.Camera.to_world.offset) // 4
_SD_PROJ_TW = (_abi.SceneDesc.proj.offset + _abi.Projector.to_world.offset) // 4
_SD_SPOT_TW = (_abi.SceneDesc.spot.offset + _abi.Spot.to_world.offset) // 4
_SD_SPOT_INT = (_abi.SceneDesc.spot.offset + _abi.Spot.intensity.offset) // 4


def set_variant(name: str) -> None:
    """Accepted for source compatibility (every example calls mi.set_variant("cuda_ad_rgb"));
    there is exactly one backend here: HIP on gfx950, fp32, AD w.r.t. the projector texture."""
    global _variant
    _variant = name


def variant() -> str:
    return _variant


# ----------------------------------------------------------------------------- value wrappers
class Float(float):
    """a scalar — or, given an array (mi.Float(pos % w), depth.py:65-66), the float array of it"""

    def __new__(cls, v=0.0):
        if isinstance(v, (torch.Tensor, np.ndarray)) and v.ndim == 0:
            return float.__new__(cls, float(v))  # (a 0-dim tensor is a scalar: mi.Float(torch.tensor(0.5)) stays usable as float(...))
        if isinstance(v, (_ArrayBase, torch.Tensor, np.ndarray, list, tuple)):
            t = v.t if isinstance(v, _ArrayBase) else torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v)
            return Float32(t.to(torch.float32))
        return float.__new__(cls, v)

    def torch(self):
        return torch.tensor([float(self)])


class _ArrayBase:
    def __init__(self, data, device=None):
        if isinstance(data, _ArrayBase):
            data = data.t
        if not isinstance(data, torch.Tensor):
            data = torch.as_tensor(np.asarray(data, dtype=np.float32))
        self.t = data if device is None else data.to(device)

    def torch(self):
        return self.t

    def __float__(self):
        if self.t.numel() != 1:
            raise TypeError(f"only a single-element array converts to a float (shape {tuple(self.t.shape)})")
        return float(self.t.reshape(-1)[0])

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, i):
        return self.t[i]

    def __iter__(self):
        return iter(self.t)

    # The reference's own call sites compute with Mitsuba's arrays (graphics/depth.py:61-69,84: `pos * scale`, `idx % w`, `idx // w`,
    # `result[~si.is_valid()] = 0`): element-wise arithmetic and masked assignment on the wrapped tensor, the result in the operand's type
    def _wrap(self, t):
        cls = type(self)
        if isinstance(getattr(cls, "t", None), property):  # (a lazily resolved image, _RenderedXf: its arithmetic yields a plain tensor wrapper)
            cls = TensorXf
        out = object.__new__(cls)
        out.t = t
        return out

    @staticmethod
    def _raw(o):
        return o.t if isinstance(o, _ArrayBase) else o

    def __setitem__(self, i, v):
        self.t[self._raw(i)] = self._raw(v)

    def __mul__(self, o):
        return self._wrap(self.t * self._raw(o))

    __rmul__ = __mul__

    def __add__(self, o):
        return self._wrap(self.t + self._raw(o))

    __radd__ = __add__

    def __sub__(self, o):
        return self._wrap(self.t - self._raw(o))

    def __rsub__(self, o):
        return self._wrap(self._raw(o) - self.t)

    def __truediv__(self, o):
        return self._wrap(self.t / self._raw(o))

    def __mod__(self, o):
        return self._wrap(self.t % self._raw(o))

    def __floordiv__(self, o):
        return self._wrap(torch.div(self.t, self._raw(o), rounding_mode="floor"))

    def __neg__(self):
        return self._wrap(-self.t)

    def __invert__(self):
        return self._wrap(~self.t)

    def __and__(self, o):
        return self._wrap(self.t & self._raw(o))

    def __or__(self, o):
        return self._wrap(self.t | self._raw(o))

    @property
    def shape(self):
        return tuple(self.t.shape)

    def __repr__(self):
        return f"{type(self).__name__}(shape={tuple(self.t.shape)})"


class Float32(_ArrayBase):
    """flat float array (mesh `vertex_positions`, scene.py:249)."""

    def __init__(self, data, device=None):
        super().__init__(data, device)
        self.t = self.t.reshape(-1)


class UInt32(_ArrayBase):
    def __init__(self, data, device=None):
        if not isinstance(data, torch.Tensor):
            data = torch.as_tensor(np.asarray(data, dtype=np.int32))
        self.t = data.reshape(-1) if device is None else data.reshape(-1).to(device)


class Color3f(_ArrayBase):
    def __init__(self, data, device=None):
        super().__init__(data, device)
        self.t = self.t.reshape(-1)[:3]

    def torch(self):
        return self.t.reshape(1, 3)


Vector3f = Color3f
Point3f = Color3f


class TensorXf(_ArrayBase):
    pass


class Vector2f(_ArrayBase):
    """[N,2] (or two broadcastable components): film positions handed to Sensor.sample_ray (depth.py:60-74)"""

    def __init__(self, x, y=None, device=None):
        if y is not None:
            xs = x.t if isinstance(x, _ArrayBase) else torch.as_tensor(x, dtype=torch.float32)
            ys = y.t if isinstance(y, _ArrayBase) else torch.as_tensor(y, dtype=torch.float32)
            xs, ys = torch.broadcast_tensors(xs.reshape(-1).float(), ys.reshape(-1).float().to(xs.device))
            x = torch.stack([xs, ys], -1)
        super().__init__(x, device)
        self.t = self.t.reshape(-1, 2)


Point2f = Vector2f


class Ray3f:
    """mi.Ray3f(o, d[, maxt]): origins / directions [N,3] (a single origin is broadcast) — depth.py:41"""

    def __init__(self, o, d, maxt=None):
        ot = o.t if isinstance(o, _ArrayBase) else torch.as_tensor(o, dtype=torch.float32)
        dt = d.t if isinstance(d, _ArrayBase) else torch.as_tensor(d, dtype=torch.float32)
        dt = dt.reshape(-1, 3).float()
        ot = ot.reshape(-1, 3).float().to(dt.device)
        self.o = ot.expand(dt.shape[0], 3) if ot.shape[0] == 1 else ot
        self.d = dt
        self.maxt = maxt  # None or [N]: hits beyond it do not count


class SurfaceInteraction3f:
    """what Scene.ray_intersect returns, as far as the reference reads it (depth.py:41-47,77-84,115-125): the distance `t` along the ray, the
    hit point `p`, validity, and `shape` — here the shape's index + 1 (0: no shape), which orders like the pointers the reference relabels"""

    def __init__(self, t, shape, prim, ray):
        self._valid = prim >= 0
        self.t = Float32(t)
        self.p = TensorXf(ray.o + t.unsqueeze(-1) * ray.d)
        self.shape = UInt32((shape + 1).to(torch.int32))
        self.prim_index = UInt32(prim)

    def is_valid(self):
        return self._valid


class _RenderedXf(TensorXf):
    """The image of a render that was issued on one of the scene's render streams (Scene._render_stream): whoever reads it first makes the
    stream that is current THEN wait for the render — `mi.render(...).torch()`, the reference's idiom, costs one event wait; a loop that
    never looks at the image (or looks later) lets consecutive renders overlap."""

    def __init__(self, img, done):
        self._img, self._done = img, done

    @property
    def t(self):
        d = self._done
        if d is not None:
            cur = torch.cuda.current_stream(self._img.device)
            cur.wait_event(d)
            self._img.record_stream(cur)  # (allocated on the render stream, consumed here)
            self._done = None
        return self._img


class _Matrix:
    def __init__(self, m):
        self.m = m

    def torch(self):
        return self.m.reshape(1, 4, 4)

    def numpy(self):
        return self.m.numpy().reshape(1, 4, 4)


class Transform4f:
    def __init__(self, m=None):
        if m is None:
            m = torch.eye(4)
        if isinstance(m, Transform4f):
            m = m._m
        if not isinstance(m, torch.Tensor):
            m = torch.as_tensor(np.asarray(m, dtype=np.float32))
        self._m = m.detach().to("cpu", torch.float32).reshape(4, 4).clone()

    @classmethod
    def _from_rows(cls, rows16):
        """from 16 host floats (row-major), copied — no tensor conversions (the native randomiser's tables)"""
        t = cls.__new__(cls)
        t._m = torch.from_numpy(np.array(rows16, dtype=np.float32).reshape(4, 4))
        return t

    @property
    def matrix(self):
        return _Matrix(self._m)

    def numpy(self):
        return self._m.numpy()

    def __repr__(self):
        return f"Transform4f({self._m.tolist()})"


class ScalarTransform3f:
    pass


def perspective_projection(film_size, crop_size, crop_offset, fov_x, near_clip, far_clip) -> Transform4f:
    """camera space -> sample space [0,1]^2 x depth [EXT Mitsuba, SURVEY App. A]:
    scale(1/rel_size) * translate(-rel_offset) * scale(-0.5, -0.5*aspect, 1) * translate(-1, -1/aspect, 0) * perspective."""
    fw, fh = float(film_size[0]), float(film_size[1])
    cw, ch = float(crop_size[0]), float(crop_size[1])
    ox, oy = float(crop_offset[0]), float(crop_offset[1])
    aspect = fw / fh
    c = 1.0 / math.tan(math.radians(float(fov_x)) * 0.5)
    P = np.array([[c, 0, 0, 0], [0, c, 0, 0], [0, 0, far_clip / (far_clip - near_clip), -near_clip * far_clip / (far_clip - near_clip)], [0, 0, 1, 0]], np.float64)
    T1 = np.eye(4)
    T1[0, 3], T1[1, 3] = -1.0, -1.0 / aspect
    S1 = np.diag([-0.5, -0.5 * aspect, 1.0, 1.0])
    T2 = np.eye(4)
    T2[0, 3], T2[1, 3] = -ox / fw, -oy / fh
    S2 = np.diag([fw / cw, fh / ch, 1.0, 1.0])
    return Transform4f((S2 @ T2 @ S1 @ T1 @ P).astype(np.float32))


# ----------------------------------------------------------------------------- scene objects
class Film:
    def __init__(self, w, h):
        self._w, self._h = int(w), int(h)

    def size(self):
        return [self._w, self._h]

    def crop_size(self):
        return [self._w, self._h]

    def crop_offset(self):
        return [0, 0]


class Sampler:
    """`independent`-sampler stand-in: holds the seed that keys the per-sample hash."""

    def __init__(self):
        self._seed, self._wavefront = 0, 0

    def seed(self, seed, wavefront_size=0):
        self._seed, self._wavefront = int(seed), int(wavefront_size)

    def wavefront_size(self):
        return self._wavefront

    # (the reference draws `sample1` from the sampler and never uses it for a perspective sensor — depth.py:72-74: wavelengths; zeros do)
    def next_1d(self):
        return Float32(torch.zeros(max(self._wavefront, 1)))

    def next_2d(self):
        return Vector2f(torch.zeros((max(self._wavefront, 1), 2)))


class Sensor:
    def __init__(self, scene, key):
        self._scene, self._key = scene, key
        self._sampler = Sampler()

    def _p(self, k):
        return self._scene._params[self._key + "." + k]

    def id(self):
        return self._key

    def film(self):
        w, h = self._scene._film_size[self._key]
        return Film(w, h)

    def sampler(self):
        return self._sampler

    def near_clip(self):
        return float(self._p("near_clip"))

    def far_clip(self):
        return float(self._p("far_clip"))

    def x_fov(self):
        return float(self._p("x_fov"))

    def world_transform(self):
        return self._p("to_world")

    def sample_ray(self, time=0, sample1=None, sample2=None, sample3=None, active=True):
        """(Ray3f, weights) for film positions sample2 in [0,1)^2 [EXT Mitsuba perspective sensor sample_ray; call sites depth.py:72-74,
        laser_estimation.py:64]: near_p = sample_to_camera (sx, sy, 0), d = to_world normalize(near_p); the ray starts ON the near plane
        (o = position + d near / d_l.z) and ends at the far plane (maxt) — so `t` of a hit is what the K7 entry points report."""
        sc = self._scene
        pos = sample2.t if isinstance(sample2, _ArrayBase) else torch.as_tensor(sample2, dtype=torch.float32)
        pos = pos.reshape(-1, 2).to(sc.device, torch.float32)
        w, h = sc._film_size[self._key]
        K = perspective_projection((w, h), (w, h), (0, 0), self.x_fov(), self.near_clip(), self.far_clip()).numpy().astype(np.float64)
        s2c = torch.as_tensor(np.linalg.inv(K), dtype=torch.float32, device=sc.device)
        tw = torch.as_tensor(sc._mat(self._key + ".to_world"), dtype=torch.float32, device=sc.device)
        q = torch.cat([pos, torch.zeros_like(pos[:, :1]), torch.ones_like(pos[:, :1])], 1) @ s2c.T
        near_p = q[:, :3] / q[:, 3:4]
        dl = near_p / near_p.norm(dim=1, keepdim=True)
        d = dl @ tw[:3, :3].T
        near_t, far_t = self.near_clip() / dl[:, 2], self.far_clip() / dl[:, 2]
        o = tw[:3, 3].unsqueeze(0) + d * near_t.unsqueeze(-1)
        return Ray3f(o, d, maxt=far_t - near_t), Color3f(torch.ones(3))


class _PinnedRing:
    """small host->device uploads that never block the host: a ring of pinned staging buffers,
    each guarded by an event so a slot is only reused after its DMA has been consumed."""

    def __init__(self, shape, depth=8):
        self.bufs = [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def upload(self, host_array, dst):
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.bufs[k].copy_(torch.as_tensor(host_array, dtype=torch.float32).reshape(self.bufs[k].shape))
        dst.copy_(self.bufs[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


class _StepPlan:
    """a compiled scene-sample push (Scene.compile_step): the ctypes tables of ffx_scene_step_h and what keeps them alive"""
    ptrs = None


class SceneParameters:
    """dict-like parameter view with Mitsuba-style keys and `.update()` (fireflies/scene.py:94,121,
    135,249,257,384).  Assignments only record the new value; `update()` pushes everything that
    changed to the device in one go (K5+K6 for geometry)."""

    def __init__(self, scene):
        self._scene = scene
        self._d = {}
        self._dirty = set()
        self._pending = None  # a callable: values of a natively pushed scene sample not yet written into the map (fireflies_amd/scene.py Scene._materialise)

    def keys(self):
        return self._d.keys()

    def items(self):
        if self._pending is not None:
            self._pending()
        return self._d.items()

    def __contains__(self, k):
        return k in self._d

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __getitem__(self, k):
        if self._pending is not None and k != "tex.data":  # (no scene sample writes the texture: mi.render's own lookup leaves the sample pending)
            self._pending()
        return self._d[k]

    def __setitem__(self, k, v):
        if k not in self._d:
            raise KeyError(f"unknown scene parameter {k!r}")
        if self._pending is not None and k != "tex.data":
            self._pending()  # (the sample's own values first: they must not overwrite this assignment later)
        self._d[k] = v
        self._dirty.add(k)
        if k == "tex.data":
            self._scene._note_texture(v)

    def _init(self, k, v):
        self._d[k] = v

    def set_mesh_pose(self, name, world, frame=None, vertices=None):
        """Fast path used by Scene.update_meshes: instead of assigning transformed vertices to
        `<mesh>.vertex_positions`, hand over the pose; the transform runs inside ffx_scene_update."""
        self._scene._set_pose(name, world, frame, vertices)
        self._dirty.add(name + ".__pose__")

    def set_mesh_pose_np(self, name, pose16, frame=None, vertices=None):
        """set_mesh_pose for a pose that already is a row of 16 host floats (Scene._apply_native: the native randomiser's tables)"""
        self._scene._set_pose_np(name, pose16, frame, vertices)
        self._dirty.add(name + ".__pose__")

    def update(self):
        if self._pending is not None:
            self._pending()
        self._scene._apply(self._dirty)
        self._dirty = set()


class Scene:
    """Device-resident scene: geometry pool + BVH blob (ops.DeviceGeometry), per-shape albedo,
    camera / projector / spot parameter blocks, the projector texture."""

    def __init__(self, data: scenes.SceneData, device="cuda", shadows=True):
        self.device = torch.device(device)
        self.data = data
        pool, tris, tri_shape, off, stride, nfr, alb = scenes.flatten(data)
        self._base_off, self._stride, self._n_frames = off.copy(), stride, nfr
        # one extra slot per mesh for caller-supplied vertices (animation functions,
        # direct `vertex_positions` assignment)
        extra, self._scratch_off = [], []
        base = pool.shape[0]
        for s, m in enumerate(data.meshes):
            self._scratch_off.append(base)
            extra.append(m.frames[0])
            base += m.frames.shape[1]
        pool = np.concatenate([pool] + extra, 0)
        # meshes that carry vertex normals are shaded with interpolated normals, re-derived from the posed vertices per update
        self.geom = ops.DeviceGeometry(pool, tris, tri_shape, off, device=self.device, smooth=[bool(getattr(m, "smooth", False)) for m in data.meshes])
        self.mesh_names = [m.name for m in data.meshes]
        self._mesh_index = {m.name: i for i, m in enumerate(data.meshes)}
        S = len(data.meshes)
        self._xforms = torch.eye(4).repeat(S, 1, 1)  # host
        self._offs = off.copy()
        # the material table the render kernels take: [S,3] Lambert albedos, or [S,16] material rows (include/ffx.h FFX_MAT_*)
        # as soon as one mesh carries a principled BSDF (`albedo` keeps its name: column 0..2 is the base colour either way)
        rows = scenes.material_rows(data)
        self._mat_stride = 3 if rows is None else scenes.MAT_STRIDE
        alb = alb if rows is None else rows
        if rows is not None and (rows[:, scenes.MAT_COLUMN["spec_trans"]] > 0).any():
            self._warn_spec_trans("scene materials")
        self._albedo_dev = torch.from_numpy(alb).to(self.device)
        self._albedo_host = alb.copy()
        self._albedo_ring = None
        self._albedo_stale = False  # the device copy lags behind _albedo_host (refreshed when somebody asks for it)
        # up to 8 material rows (42 albedos) travel inside the scene description as kernel arguments: a randomisation of the
        # material then costs no host-to-device copy at all (it was the one copyBuffer launch of every step)
        self._mats_in_sd = alb.size <= 128 and os.environ.get("FFX_HOST_MATERIALS", "1") != "0"
        self._apex_ahead = os.environ.get("FFX_APEX_AHEAD", "1") != "0"  # apex records written with the re-fit instead of in front of the render
        # Two render streams, used in turn (FFX_RENDER_STREAMS=1: the caller's stream, as before): consecutive renders are independent of
        # each other — each reads its own blob copy, the texture and kernel arguments — and back to back on ONE stream they leave the tail
        # of every launch (the last of 262 144 one-wave workgroups draining) and the dependency gap behind it unused: tools/overlapprobe.py,
        # 519.2 us per render on one stream, 505.5 on two.  What a render depends on is waited for explicitly (the texture's assignment,
        # the re-fit of its blob); the image is handed out as _RenderedXf.
        self._render_streams = None
        if self.device.type == "cuda" and os.environ.get("FFX_RENDER_STREAMS", "2") != "1":
            self._render_streams = list(ops.shared_streams(self.device, "render", 2))  # (the process's pair on this device: ops.shared_streams)
            self._render_done = [[torch.cuda.Event() for _ in range(4)] for _ in range(2)]
        self._render_turn = 0
        # which path each mi.render of this scene took (bench.py prints it): "two_stream" = beside the previous render on the scene's
        # render streams, "caller_stream" = the plain forward on the caller's stream (a texture written in place since its assignment,
        # device material tables, base-colour textures, FFX_RENDER_STREAMS=1), "autograd" = through functional.render (tex.data requires grad)
        self.render_paths = {"two_stream": 0, "caller_stream": 0, "autograd": 0}
        # which way each geometry push went: "native" = one ffx_scene_step_h call for the whole sample (step_native), "python" = params.update()
        self.update_paths = {"native": 0, "python": 0}
        self.update_fallbacks = {}  # why a sample of a compiled configuration took the Python path after all (reason -> count)
        self._tex_src = self._tex_ready = self._tex_private = None
        self._tex_ver = -1
        # texture-valued base colours (`<mat>.brdf_0.base_color.data`): device tensors [h,w,3] + the texture coordinates of every
        # leaf slot's three corners (static: the slot order is the tree's)
        self._base_tex = [(name, torch.from_numpy(t).to(self.device).contiguous()) for name, t in scenes.base_textures(data)]
        self._slot_uv = None
        if self._base_tex:
            info = self.geom.info
            order = self.geom._blobs[0][int(info.off_order): int(info.off_order) + 4 * int(info.n_tris)].cpu().numpy().view(np.int32)
            self._slot_uv = torch.from_numpy(scenes.slot_uv_table(order, tris, tri_shape, data.meshes)).to(self.device).contiguous()
        self.shadows = shadows
        self._low_spp = False  # (note_spp)
        self.tex_color = (0.0, 1.0, 0.0)
        # the camera film's reconstruction filter: "box" or "gaussian" / ("gaussian", stddev).  A scene file gets what it declares — and
        # hdrfilm's default, the gaussian, when it declares none, like every scene the reference loads (loaders.load_xml puts it into
        # data.notes); procedural scenes (scenes.py) default to the box.  Assigning the attribute re-derives the scene description.
        self._rfilter = None
        rf = (getattr(data, "notes", None) or {}).get("rfilter")
        if rf == "gaussian":
            self._rfilter = ("gaussian", float(data.notes.get("rfilter_stddev", 0.5)))
        self._film_size = {}
        self._params = SceneParameters(self)
        self._build_params()
        self._sensors = [Sensor(self, data.camera.name)]
        if data.projector is not None:
            self._sensors.append(Sensor(self, data.projector.name))
        self._sd_cache = None
        self._sd_templates = {}  # tex_channels -> (key of what else it depends on, finished description): scene_desc() patches the per-step fields into a copy
        self._key_plan = {}      # parameter key -> what _apply does for it (worked out the first time the key is seen)
        self._key_dyn = {}       # parameter key -> is it one of the per-step fields scene_desc() patches?

    # ------------------------------------------------------------------ parameters
    def _build_params(self):
        p, d = self._params, self.data
        for m in d.meshes:
            p._init(m.name + ".vertex_positions", Float32(torch.from_numpy(m.frames[0].reshape(-1).copy())))
            p._init(m.name + ".faces", UInt32(m.tris))
            p._init(m.name + ".vertex_count", int(m.frames.shape[1]))
            p._init(m.name + ".face_count", int(m.tris.shape[0]))
        mats = {}
        for m in d.meshes:
            mats.setdefault(m.material, []).append(m)
        self._material_meshes = {k: [self._mesh_index[x.name] for x in v] for k, v in mats.items()}
        # principled materials expose the plugin's parameters under the names Mitsuba gives them inside the exporter's
        # `twosided` wrapper (main.py:97-107: "brdf_0.roughness.value", ..., and "brdf_0.specular" without ".value");
        # diffuse materials keep base_color / specular / roughness so that the reference's scripts run on them
        self._material_principled = {k: v[0].bsdf is not None for k, v in mats.items()}
        for name, t in self._base_tex:  # (Mitsuba exposes a bitmap-valued base colour as `.data`, a constant one as `.value`)
            p._init(name + ".brdf_0.base_color.data", TensorXf(t))
        for mat, ms in mats.items():
            p._init(mat + ".brdf_0.base_color.value", Color3f(torch.tensor(ms[0].albedo, dtype=torch.float32)))
            b = dict(scenes.PRINCIPLED_DEFAULTS)
            b.update(ms[0].bsdf or {})
            p._init(mat + ".brdf_0.specular", Float(b["specular"]))
            p._init(mat + ".brdf_0.roughness.value", Float(b["roughness"]))
            if ms[0].bsdf is not None:
                for k in scenes.PRINCIPLED_DEFAULTS:
                    if k not in ("specular", "roughness"):
                        p._init(mat + f".brdf_0.{k}.value", Float(b[k]))
                p._init(mat + ".brdf_0.eta", Float(scenes.specular_to_eta(b["specular"]) if "eta" not in (ms[0].bsdf or {}) else ms[0].bsdf["eta"]))
        for s in [d.camera] + ([d.projector] if d.projector is not None else []):
            p._init(s.name + ".to_world", Transform4f(s.to_world))
            p._init(s.name + ".x_fov", Float(s.fov_x))
            p._init(s.name + ".near_clip", Float(s.near))
            p._init(s.name + ".far_clip", Float(s.far))
            self._film_size[s.name] = (s.width, s.height)
        if d.projector is not None:
            p._init("Projector.to_world", Transform4f(d.projector.to_world))
            p._init("Projector.scale", Float(d.projector_scale))
            tex = torch.zeros((d.projector.height, d.projector.width, 3), device=self.device)
            p._init("tex.data", TensorXf(tex))
        if d.spot is not None:
            s = d.spot
            p._init(s.name + ".to_world", Transform4f(s.to_world))
            p._init(s.name + ".intensity.value", Color3f(torch.tensor(s.intensity, dtype=torch.float32)))
            p._init(s.name + ".cutoff_angle", Float(s.cutoff_angle))
            p._init(s.name + ".beam_width", Float(s.beam_width))

    def sensors(self):
        return self._sensors

    def ray_intersect(self, ray, active=True):
        """scene.ray_intersect(mi.Ray3f(o, d)) (depth.py:41,77,115,157): the closest hit of every ray against the current pose
        (ffx_trace_rays, K7) — `t`, `p`, `is_valid()`, `shape` (index + 1; 0 without a hit)"""
        o, d = ray.o.to(self.device).contiguous(), ray.d.to(self.device).contiguous()
        t, shape, prim = self.geom.trace_rays(o, d)
        if ray.maxt is not None:
            far = torch.as_tensor(ray.maxt, dtype=torch.float32, device=self.device).reshape(-1)
            miss = t > far
            prim = torch.where(miss, torch.full_like(prim, -1), prim)
            shape = torch.where(miss, torch.full_like(shape, -1), shape)
            t = torch.where(miss, torch.zeros_like(t), t)
        return SurfaceInteraction3f(t, shape, prim, Ray3f(o, d))

    def shapes(self):
        return list(self.mesh_names)

    def _set_pose(self, name, world, frame, vertices):
        i = self._mesh_index[name]
        w = world.detach().to("cpu", torch.float32) if isinstance(world, torch.Tensor) else torch.as_tensor(np.asarray(world, np.float32))
        self._xforms[i] = w.reshape(4, 4)
        if vertices is not None:
            V = int(self._stride[i])
            v = vertices.detach().to(self.device, torch.float32).reshape(-1, 3)
            if v.shape[0] != V:
                raise ValueError(f"{name}: expected {V} vertices, got {v.shape[0]}")
            o = self._scratch_off[i]
            self.geom.write_verts(o, v)  # ordered against the refits on the side stream
            self._offs[i] = o
        elif frame is not None:
            f = int(frame)
            if not (0 <= f < int(self._n_frames[i])):
                raise IndexError(f"{name}: frame {f} out of range [0, {int(self._n_frames[i])})")
            self._offs[i] = self._base_off[i] + f * int(self._stride[i])

    def _set_pose_np(self, name, pose16, frame, vertices):
        i = self._mesh_index[name]
        xn = getattr(self, "_xforms_np", None)
        if xn is None:
            xn = self._xforms_np = self._xforms.numpy()  # (shares the host tensor's memory; _xforms is only ever written in place)
        xn[i] = pose16.reshape(4, 4)
        if vertices is not None:
            self._set_pose(name, self._xforms[i], None, vertices)
        elif frame is not None:
            f = int(frame)
            if not (0 <= f < int(self._n_frames[i])):
                raise IndexError(f"{name}: frame {f} out of range [0, {int(self._n_frames[i])})")
            self._offs[i] = self._base_off[i] + f * int(self._stride[i])

    def _apply(self, dirty):
        geom_dirty = False
        albedo_dirty = False
        plan = self._key_plan
        # (`specular` last: Mitsuba's principled plugin re-derives eta from it whenever it is among the updated keys [EXT principled.cpp
        # parameters_changed], so it overrides an `eta` assigned in the same update — and the result must not depend on the order a set is walked in)
        late = []
        for keys in (dirty, late):
            for k in keys:
                kp = plan.get(k)
                if kp is not None:
                    # (a key seen before: what it means was worked out then — 19 keys per randomisation went through the string handling below, 25 us)
                    if kp is _POSE_KEY:
                        geom_dirty = True
                    elif kp is not _PLAIN_KEY:
                        col, rows, conv, eta_key, warn_st = kp
                        if conv and keys is dirty:
                            late.append(k)
                            continue
                        v = self._params._d[k]
                        v = float(v) if isinstance(v, float) else float((v.t if isinstance(v, _ArrayBase) else torch.as_tensor(v, dtype=torch.float32)).reshape(-1)[0])
                        if conv:  # `specular`: the plugin re-derives eta from it
                            v = scenes.specular_to_eta(v)
                            self._params._d[eta_key] = Float(v)
                        if warn_st and v > 0.0:
                            self._warn_spec_trans(k)
                        for i in rows:
                            self._albedo_host[i, col] = v
                        albedo_dirty = True
                    continue
                base, _, rest = k.partition(".")
                if rest == "__pose__":
                    geom_dirty = True
                    plan[k] = _POSE_KEY
                elif rest == "vertex_positions" and base in self._mesh_index:
                    # generic Mitsuba-style path: the caller transformed the vertices itself
                    v = self._params._d[k]
                    v = v.t if isinstance(v, _ArrayBase) else torch.as_tensor(v)
                    self._set_pose(base, torch.eye(4), None, v)
                    geom_dirty = True
                elif rest.startswith("brdf_0.") and rest not in ("brdf_0.base_color.value", "brdf_0.base_color.data") and base in self._material_meshes:
                    # principled-BSDF parameters (specular, roughness, clearcoat, ...; the reference randomises them:
                    # main.py:97-107, examples/vocalfold_scene.py:93)
                    name = rest[len("brdf_0."):]
                    name = name[:-len(".value")] if name.endswith(".value") else name
                    if name == "specular" and keys is dirty:
                        late.append(k)
                        continue
                    if not self._material_principled.get(base, False):
                        # a diffuse material has no such parameters in Mitsuba; accepted so that the scripts run — say so once
                        if not getattr(self, "_warned_bsdf", False):
                            import warnings

                            warnings.warn(f"{k}: material {base!r} is diffuse — principled-BSDF parameters do not affect it "
                                          "(declare it principled: scenes.MeshData(bsdf={...}) or <bsdf type=\"principled\">); "
                                          "further assignments of such parameters are not reported", stacklevel=4)
                            self._warned_bsdf = True
                        plan[k] = _PLAIN_KEY  # (no part of anything the device sees)
                        continue
                    v = self._params._d[k]
                    v = float(v) if isinstance(v, float) else float((v.t if isinstance(v, _ArrayBase) else torch.as_tensor(v, dtype=torch.float32)).reshape(-1)[0])
                    if name == "specular":  # the plugin re-derives eta from it (principled.cpp parameters_changed)
                        col, v = scenes.MAT_COLUMN["eta"], scenes.specular_to_eta(v)
                        self._params._d[base + ".brdf_0.eta"] = Float(v)
                    elif name in scenes.MAT_COLUMN and name != "model":
                        col = scenes.MAT_COLUMN[name]
                    else:
                        raise KeyError(f"{k}: not a parameter of the principled BSDF")
                    if name == "spec_trans" and v > 0.0:
                        self._warn_spec_trans(k)
                    for i in self._material_meshes[base]:
                        self._albedo_host[i, col] = v
                    albedo_dirty = True
                    plan[k] = (col, tuple(self._material_meshes[base]), name == "specular", base + ".brdf_0.eta", name == "spec_trans")
                elif rest == "brdf_0.base_color.data" and any(n == base for n, _ in self._base_tex):
                    # a new base-colour texture (main.py:147-153 assigns one per iteration, through numpy): any resolution, [h,w,3]
                    v = self._params._d[k]
                    v = v.t if isinstance(v, _ArrayBase) else torch.as_tensor(np.asarray(v, np.float32))
                    if v.dim() != 3 or v.shape[-1] != 3:
                        raise ValueError(f"{k}: expected a [height, width, 3] texture, got {tuple(v.shape)}")
                    v = v.detach().to(self.device, torch.float32).contiguous()
                    self._base_tex = [(n, v if n == base else t) for n, t in self._base_tex]
                    self._params._d[k] = TensorXf(v)
                elif rest == "brdf_0.base_color.value" and base in self._material_meshes:
                    c = self._params._d[k]
                    c = (c.t if isinstance(c, _ArrayBase) else torch.as_tensor(c, dtype=torch.float32)).reshape(-1)[:3]
                    c = c.detach().cpu().numpy()
                    for i in self._material_meshes[base]:
                        self._albedo_host[i, :3] = c
                    albedo_dirty = True
                else:
                    if not (rest.startswith("brdf_0.") or rest == "vertex_positions"):
                        plan[k] = _PLAIN_KEY  # (an emitter / sensor / texture parameter: read when the scene description is formed)
        if albedo_dirty:
            if self._mats_in_sd:
                self._albedo_stale = True  # the next scene_desc() carries the new rows; the device tensor is refreshed on demand
            else:
                self._upload_albedo()
        ch = self._sd_cache[0] if self._sd_cache is not None else 3
        self._sd_cache = None
        kd = self._key_dyn
        for k in dirty:
            # (the fields a randomisation writes every step — poses, emitter poses, intensity and cone, material values — are patched into a copy of the
            # finished description; anything else — a field of view, a clip plane, the projector's scale, a new base-colour texture — rebuilds it)
            dyn = kd.get(k)
            if dyn is None:
                dyn = kd[k] = (k.endswith(".__pose__") or k.endswith(".to_world") or k.endswith(".intensity.value") or k.endswith(".vertex_positions") or k == "tex.data"
                               or k.endswith(".cutoff_angle") or k.endswith(".beam_width") or (".brdf_0." in k and not k.endswith(".data")))
            if not dyn:
                self._sd_templates = {}
                break
        else:
            # the templates themselves keep up with assignments of the description's per-step fields — a pose of the camera, the projector or the
            # spot, the spot's intensity or cone: the native push (step_native) starts from a template and only writes the fields ITS plan has ops
            # for, i.e. those of randomised entities; a caller who moves a fixed camera between samples would have rendered every natively pushed
            # sample from the old pose (round-5 advisor; the Python path reads the map every time)
            if self._sd_templates and any(k.endswith(".to_world") or k.endswith(".intensity.value") or k.endswith(".cutoff_angle") or k.endswith(".beam_width") for k in dirty):
                for _, keep in self._sd_templates.values():
                    self._patch_sd(keep)
        if geom_dirty:
            # the renders of this pose will come from the camera / emitter positions the parameters hold NOW (every assignment of
            # the update has been applied above): their apex records are written behind the re-fit, on its side stream, and the
            # render launches without a pre-pass (ops.DeviceGeometry.update; the description is built here instead of in render())
            self.update_paths["python"] += 1
            self.geom.update(self._xforms, self._offs, apex_sd=self.scene_desc(tex_channels=ch) if self._apex_ahead else None)

    # ------------------------------------------------------------------ the native params.update() (include/ffx.h ffx_scene_step_h)
    # A scene sample drawn by the native randomiser (ffx_scene_randomize_h) reaches the device in ONE call: its key writes — poses into
    # the description's to_world blocks, the spot's attributes, material attributes into the material rows, mesh poses into the transform
    # table — are compiled once into a table of ops, from what _apply / scene_desc() LEARNED about each key when the Python path pushed a
    # sample (`_key_plan`, `_key_dyn`): the native path can only do what the Python path was seen doing, and anything it was not — a key
    # that rebuilds the description, caller-supplied vertices, a device material table — keeps the Python path (fireflies_amd/scene.py).
    def compile_step(self, poses, values, meshes, n_draws, n_ents):
        """poses: [(entity row, parameter key)], values: [(draw row, values per draw, repeat, parameter key)], meshes: [(entity row, mesh
        name)] — the writes of Scene._apply_native.  -> the compiled plan, or None (not compilable, or not yet: a key never pushed before)"""
        g = self.geom
        if not (self._apex_ahead and g._async and g.n_shapes <= 32 and self.device.type == "cuda") or ops._lane_kernels():
            return None
        d, stride = self.data, self._mat_stride
        sd_pose = {d.camera.name + ".to_world": _SD_CAM_TW}
        if d.projector is not None:
            sd_pose["Projector.to_world"] = _SD_PROJ_TW
        spot = d.spot.name if d.spot is not None else None
        if spot is not None:
            sd_pose[spot + ".to_world"] = _SD_SPOT_TW
        out, eta, late, warn = [], [], [], []
        for row, key in poses:
            if self._key_dyn.get(key) is not True:
                return None
            dst = sd_pose.get(key)
            if dst is not None:  # (any other entity's pose is no part of the description)
                out.append((_abi.STEP_POSE_SD, row, 0, dst, 0, 0))
        for drow, n, rep, key in values:
            if self._key_dyn.get(key) is not True:
                return None
            comp = (lambda j: j) if (rep == 1 and n == 3) else (lambda j: 0)
            kp = self._key_plan.get(key)
            if kp is None:
                base, _, rest = key.partition(".")
                if rest == "brdf_0.base_color.value" and base in self._material_meshes and (n == 3 or rep == 3):
                    for i in self._material_meshes[base]:
                        out += [(_abi.STEP_VALUE_MAT, drow, comp(j), i * stride + j, 0, 0) for j in range(3)]
                    continue
                return None
            if kp is _POSE_KEY:
                return None
            if kp is _PLAIN_KEY:
                if spot is not None and key == spot + ".intensity.value":
                    if not (n == 3 or rep == 3):
                        return None
                    out += [(_abi.STEP_VALUE_SD, drow, comp(j), _SD_SPOT_INT + j, 0, 0) for j in range(3)]
                elif spot is not None and key in (spot + ".cutoff_angle", spot + ".beam_width"):
                    if n != 1 or rep != 1:
                        return None
                    out.append((_abi.STEP_VALUE_SD, drow, 0, _SD_SPOT_INT + (3 if key.endswith("cutoff_angle") else 4), 0, 0))
                continue  # (anything else that is plain and per-step is no part of the description)
            col, rows, conv, eta_key, warn_st = kp
            if warn_st:
                warn.append((key, drow))  # (a value > 0 is reported once per scene: step_native looks)
            (late if conv else out).extend((_abi.STEP_VALUE_MAT, drow, 0, i * stride + col, 1 if conv else 0, 0) for i in rows)
            if conv:
                eta.append((eta_key, drow))
        out += late  # (`specular` -> eta after everything else, as in _apply)
        has_mat = any(o[0] == _abi.STEP_VALUE_MAT for o in out)
        if has_mat and not self._mats_in_sd:
            return None  # (a material table on the device is uploaded by the Python path)
        for row, name in meshes:
            out.append((_abi.STEP_MESH, row, 0, self._mesh_index[name], 0, 0))
        sp = _StepPlan()
        sp.ops = (_abi.StepOp * max(len(out), 1))()
        for o, t in zip(sp.ops, out):
            o.kind, o.src, o.comp, o.dst, o.conv, o.mode = t
        S = g.n_shapes
        sp.tabs = [np.ascontiguousarray(a, np.int32) for a in (self._base_off, self._stride, self._n_frames)]
        sp.plan = _abi.StepPlan()
        sp.plan.ops, sp.plan.n_ops, sp.plan.n_shapes, sp.plan.n_draws, sp.plan.n_ents = sp.ops, len(out), S, int(n_draws), int(n_ents)
        sp.plan.frame_base, sp.plan.frame_stride, sp.plan.n_frames = (t.ctypes.data_as(C.POINTER(C.c_int32)) for t in sp.tabs)
        sp.plan.n_mat_floats = int(self._albedo_host.size) if self._mats_in_sd else 0
        sp.frames = (C.c_int32 * S)(*([-1] * S))
        sp.mesh_shapes = [self._mesh_index[name] for _, name in meshes]
        sp.eta, sp.has_mat, sp.warn = eta, has_mat, warn
        sp.geoms = []
        for b in g._blobs:
            gs = _abi.StepGeom()
            gs.bvh, gs.info = b.data_ptr(), C.pointer(g.info)
            gs.src_verts, gs.tris, gs.tri_shape = g.src_verts.data_ptr(), g.tris.data_ptr(), g.tri_shape.data_ptr()
            gs.smooth = C.pointer(g._smooth[0]) if g._smooth is not None else None
            sp.geoms.append(gs)
        sp.pool = g.src_verts  # (the tensors whose addresses the blocks hold)
        sp.fn = ops.api().lib.ffx_scene_step_h
        return sp

    def step_native(self, sp, values, chain, unc, frames):
        """one scene sample through ffx_scene_step_h.  values [n_draws,4], chain / unc [n_ents,16]: rows of the native randomiser's tables (float32,
        contiguous); frames: per posed mesh of the plan the pool frame of this sample.  False: not this time (the caller takes the Python path)"""
        p, fb = self._params, self.update_fallbacks
        if p._dirty and not (len(p._dirty) == 1 and "tex.data" in p._dirty):
            fb["caller's own assignments pending"] = fb.get("caller's own assignments pending", 0) + 1
            return False  # (params.update() applies them in its order)
        ch = self._sd_cache[0] if self._sd_cache is not None else 3
        tm = self._sd_templates.get(ch)
        if tm is None or tm[0] != (self._shadows_word(), tuple(self.tex_color), self._mat_stride, self._mats_in_sd, self._rfilter, self._slot_uv.data_ptr() if self._slot_uv is not None else 0):
            fb["no description template yet"] = fb.get("no description template yet", 0) + 1
            return False
        g = self.geom
        if g.src_verts is not sp.pool or (g.device.index is not None and torch._C._cuda_getDevice() != g.device.index):
            return False  # (a scene on another device than the current one: DeviceGeometry._call switches for the Python path's launches)
        fr = sp.frames
        for s, f in zip(sp.mesh_shapes, frames):
            fr[s] = f
        sd = _abi.SceneDesc()
        sd._frozen = True  # (never written again once the call has filled it in: ops.apex_key may remember its key on it)
        ptrs = sp.ptrs
        if ptrs is None or ptrs[3] is not self._albedo_host:  # (host tables only ever written in place: their addresses are taken once)
            ptrs = sp.ptrs = (self._xforms.data_ptr(), self._offs.ctypes.data, self._albedo_host.ctypes.data if sp.plan.n_mat_floats else None, self._albedo_host)

        # what the sample's description starts from: the LAST finished description when there is one — every field this plan has no op for then keeps
        # the value the scene was last pushed with, as on the Python path, where such a field is read from the parameter map (a second Scene over the
        # same map randomises other entities than the first: started from the template, its samples carried the spot intensity of the day the
        # template was built — found by the round-6 action fuzz) —, else the template
        base = self._sd_cache[1] if self._sd_cache is not None else tm[1]
        if base.shadows != tm[1].shadows:  # (note_spp changed the regime since that description was made: the new sample carries the new word)
            nb = _abi.SceneDesc()
            C.memmove(C.addressof(nb), C.addressof(base), C.sizeof(nb))
            nb.shadows = tm[1].shadows
            base = nb

        def launch(i, stream, prepare):
            rc = sp.fn(sp.plan, values.ctypes.data, chain.ctypes.data, unc.ctypes.data, fr, base, sd, ptrs[2], ptrs[0], ptrs[1], sp.geoms[i], prepare, stream)
            if rc != 0:
                ops.api().check(rc, "ffx_scene_step_h")
            return sd

        g.update_native(launch, self._offs)
        self.update_paths["native"] += 1
        if sp.warn and not getattr(self, "_warned_spec_trans", False):
            for key, drow in sp.warn:
                if values[drow, 0] > 0.0:
                    self._warn_spec_trans(key)
        p._dirty = set()
        if sp.has_mat:
            self._albedo_stale = True
        self._sd_cache = (ch, sd)
        return True

    def _note_texture(self, v):
        """`params["tex.data"] = v`: from here on the texture is this tensor as the current stream leaves it — what a render stream waits for"""
        t = v.t if isinstance(v, _ArrayBase) else v
        self._tex_src = self._tex_private = None
        mine = self.device.index if self.device.index is not None else torch.cuda.current_device() if self.device.type == "cuda" else None
        if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index == mine and not t.requires_grad and self._render_streams is not None:
            # the render streams read a PRIVATE copy (as Mitsuba's scene owns its copy of an assigned tensor): the caller may write into
            # its tensor right after a render whose image it has not looked at yet — that render must not see the write
            cur = torch.cuda.current_stream(t.device)
            priv = t.detach().to(torch.float32).contiguous().clone()
            for rs in self._render_streams:
                priv.record_stream(rs)
            if self._tex_ready is None:
                self._tex_ready = torch.cuda.Event()
            self._tex_ready.record(cur)
            self._tex_src, self._tex_ver, self._tex_private = t, t._version, priv

    def _render_stream(self, tex, mats):
        """(stream, completion event) for a render of `tex` that can run beside the previous one — or None: the caller's stream.  Only when
        everything the render reads is accounted for: the texture is the tensor that was assigned, untouched since (its version counter), the
        materials travel as kernel arguments, no base-colour textures."""
        if self._render_streams is None or tex is not self._tex_src or tex._version != self._tex_ver or mats is not None or self._base_tex or not self.geom._async:
            return None  # (a texture written in place since its assignment: the live tensor, on the caller's stream, as before)
        i = self._render_turn & 1
        self._render_turn += 1
        rs = self._render_streams[i]
        rs.wait_event(self._tex_ready)
        return rs, self._render_done[i][(self._render_turn >> 1) & 3], self._tex_private

    def _upload_albedo(self):
        if self._albedo_ring is None:
            self._albedo_ring = _PinnedRing(tuple(self._albedo_host.shape))
        self._albedo_ring.upload(self._albedo_host, self._albedo_dev)
        self._albedo_stale = False

    @property
    def albedo(self):
        """the material table on the device ([S,3] albedos or [S,16] material rows).  With the rows inside the scene description
        (the default for small tables) the render calls do not read it; it is brought up to date whenever it is asked for."""
        if self._albedo_stale:
            self._upload_albedo()
        return self._albedo_dev

    def materials_arg(self, sd):
        """what to pass as the render calls' material table for `sd`: None when sd carries the rows itself"""
        return None if sd.n_mat_h > 0 else self.albedo

    def _warn_spec_trans(self, what):
        """once per scene: the transmission lobe of `spec_trans` is not evaluated (main.py:105 randomises it 0 .. 0.4)"""
        if not getattr(self, "_warned_spec_trans", False):
            import warnings

            warnings.warn(f"{what}: spec_trans > 0 — the principled BSDF's TRANSMISSION lobe is not evaluated here; spec_trans only scales the diffuse "
                          "lobe by (1 - spec_trans).  That is what Mitsuba does for the reference's materials too: they are principled BSDFs nested in "
                          "`twosided` (`<mat>.brdf_0.*`), which only accepts a BSDF without a transmission component, so the plugin's lobe is switched "
                          "off at load time (DESIGN.md 8).  A principled BSDF used WITHOUT `twosided` and lit from behind would differ.  "
                          "Not reported again for this scene.", stacklevel=5)
            self._warned_spec_trans = True

    # ------------------------------------------------------------------ render-time blocks
    def _mat(self, key):
        v = self._params[key]
        return v.numpy() if isinstance(v, Transform4f) else np.asarray(v, np.float32).reshape(4, 4)

    @property
    def rfilter(self):
        return "box" if self._rfilter is None else self._rfilter

    @rfilter.setter
    def rfilter(self, value):
        scene_desc.set_rfilter(_abi.SceneDesc(), value)  # (validates)
        self._rfilter = None if value in (None, "box") else (value if not isinstance(value, str) else (value, 0.5))
        self._sd_cache = None
        self._sd_templates = {}

    def _shadows_word(self):
        """include/ffx.h ffx_scene_desc.shadows: on / off, plus the hint that this scene's renders are short (note_spp)"""
        return (0 if not self.shadows else (3 if self._low_spp else 1)) | (4 if getattr(self, "_cache_dense", False) else 0)

    def set_cache_dense(self, dense=True):
        """FFX_SHADOWS_CACHE_DENSE for the descriptions to come: the filtered film's adjoint cache with a block for every pass of every pixel (it cannot
        overflow) instead of the arena's share — PatternOptimizer's answer to an overflow.  Like note_spp: templates and the finished description take
        the new word, nothing is rebuilt."""
        dense = bool(dense)
        if dense != getattr(self, "_cache_dense", False):
            self._cache_dense = dense
            bit = 4 if dense else 0
            for ch, (tkey, keep) in list(self._sd_templates.items()):
                keep.shadows = (int(keep.shadows) & ~4) | bit
                self._sd_templates[ch] = ((int(keep.shadows),) + tuple(tkey[1:]), keep)
            if self._sd_cache is not None:  # (only this bit: the finished description keeps the pre-pass hint it was prepared with)
                self._sd_cache[1].shadows = (int(self._sd_cache[1].shadows) & ~4) | bit

    def note_spp(self, spp):
        """called by whoever renders this scene with `spp` samples per pixel: below 33 the pre-pass of the NEXT poses leaves the emitters' envelopes
        out and bins the spot on a coarser grid (FFX_SHADOWS_PLAIN: a short render waits for the pre-pass chain; a long one hides it and runs
        15 % faster with the envelopes).  Only poses pushed from now on are affected: the finished description of the current pose — the one its
        pre-pass was run with — stays as it is, the templates take the new word (a loop that draws its sample count per render, main.py:145,
        changes regime every few samples: nothing is rebuilt, a pose is merely prepared with the hint of the render before it)."""
        low = int(spp) < 33
        if low != self._low_spp:
            self._low_spp = low
            w = self._shadows_word()
            for ch, (tkey, keep) in list(self._sd_templates.items()):
                keep.shadows = w
                self._sd_templates[ch] = ((w,) + tuple(tkey[1:]), keep)

    def _patch_sd(self, sd):
        """the per-step fields of a finished description from the parameter map as it is NOW: the three poses, the spot's intensity and cone"""
        f = np.frombuffer(sd, dtype=np.float32)
        d, p = self.data, self._params
        f[_SD_CAM_TW:_SD_CAM_TW + 16] = self._mat(d.camera.name + ".to_world").reshape(-1)
        if d.projector is not None:
            f[_SD_PROJ_TW:_SD_PROJ_TW + 16] = self._mat("Projector.to_world").reshape(-1)
        if d.spot is not None:
            f[_SD_SPOT_TW:_SD_SPOT_TW + 16] = self._mat(d.spot.name + ".to_world").reshape(-1)
            inten = p[d.spot.name + ".intensity.value"]
            f[_SD_SPOT_INT:_SD_SPOT_INT + 3] = inten.t.reshape(-1)[:3].tolist() if isinstance(inten, _ArrayBase) else [float(v) for v in inten]
            f[_SD_SPOT_INT + 3], f[_SD_SPOT_INT + 4] = float(p[d.spot.name + ".cutoff_angle"]), float(p[d.spot.name + ".beam_width"])

    def scene_desc(self, tex_channels=3):
        if self._sd_cache is not None and self._sd_cache[0] == tex_channels:
            return self._sd_cache[1]
        tkey = (self._shadows_word(), tuple(self.tex_color), self._mat_stride, self._mats_in_sd, self._rfilter, self._slot_uv.data_ptr() if self._slot_uv is not None else 0)
        tm = self._sd_templates.get(tex_channels)
        if tm is not None and tm[0] == tkey:
            # a copy of the finished description with this step's fields written over it (22 -> 6 us per step): the three poses, the spot's
            # intensity, the material rows — everything else was read when the template was built and has not been assigned since (_apply)
            sd = _abi.SceneDesc()
            C.memmove(C.addressof(sd), C.addressof(tm[1]), C.sizeof(sd))
            self._patch_sd(sd)
            if self._mats_in_sd:
                scene_desc.set_host_materials(sd, self._albedo_host)
            sd._frozen = True  # (never written again: ops.apex_key may remember its key on it)
            self._sd_cache = (tex_channels, sd)
            return sd
        d, p = self.data, self._params
        cam = d.camera
        sensor = scenes.SensorData(cam.name, self._mat(cam.name + ".to_world"), float(p[cam.name + ".x_fov"]), float(p[cam.name + ".near_clip"]),
                                   float(p[cam.name + ".far_clip"]), cam.width, cam.height)
        proj = None
        if d.projector is not None:
            pr = d.projector
            proj = scenes.SensorData(pr.name, self._mat("Projector.to_world"), float(p[pr.name + ".x_fov"]), float(p[pr.name + ".near_clip"]),
                                     float(p[pr.name + ".far_clip"]), pr.width, pr.height)
        spot = None
        if d.spot is not None:
            s = d.spot
            inten = p[s.name + ".intensity.value"]
            inten = inten.t.reshape(-1).tolist() if isinstance(inten, _ArrayBase) else list(inten)
            spot = scenes.SpotData(s.name, self._mat(s.name + ".to_world"), tuple(float(v) for v in inten), float(p[s.name + ".cutoff_angle"]),
                                   float(p[s.name + ".beam_width"]))
        tmp = scenes.SceneData(d.meshes, sensor, proj, spot, float(p["Projector.scale"]) if proj is not None else 1.0)
        btex = [(t.data_ptr(), t.shape[1], t.shape[0]) for _, t in self._base_tex] or None
        sd = scene_desc.scene_desc(tmp, tex_channels=tex_channels, color=self.tex_color, shadows=self._shadows_word(), mat_stride=self._mat_stride, base_tex=btex,
                                   slot_uv=self._slot_uv.data_ptr() if self._slot_uv is not None else None,
                                   host_mats=self._albedo_host if self._mats_in_sd else None, rfilter=self._rfilter)
        sd._frozen = True
        self._sd_cache = (tex_channels, sd)
        if not self._base_tex:  # (base-colour textures are re-assigned per iteration by the dataset loop: their addresses stay out of a template)
            keep = _abi.SceneDesc()
            C.memmove(C.addressof(keep), C.addressof(sd), C.sizeof(sd))
            self._sd_templates[tex_channels] = (tkey, keep)
        return sd

    def camera_struct(self, index=0):
        if index == 0 and self._sd_cache is not None:
            # the finished description of this pose already holds the camera (the dataset loop asks twice per sample, depth.py's queries: 85 us
            # each through the sensor's parameters).  Its block is built by another routine (scenes.perspective_projection) than the one below:
            # taken only after the two have been seen to agree byte for byte for this scene's static camera parameters (= for this template)
            ch, sd = self._sd_cache
            tm = self._sd_templates.get(ch)
            if tm is not None:
                chk = getattr(self, "_cam_check", None)
                if chk is None or chk[0] is not tm:
                    self._cam_check = chk = (tm, None)
                if chk[1] is None:
                    slow = self._camera_struct_slow(0)
                    self._cam_check = chk = (tm, C.string_at(C.addressof(slow), C.sizeof(slow)) == C.string_at(C.addressof(sd.cam), C.sizeof(sd.cam)))
                    return slow
                if chk[1]:
                    c = _abi.Camera()
                    C.memmove(C.addressof(c), C.addressof(sd.cam), C.sizeof(c))
                    return c
        return self._camera_struct_slow(index)

    def _camera_struct_slow(self, index=0):
        s = self._sensors[index]
        w, h = self._film_size[s._key]
        K = perspective_projection((w, h), (w, h), (0, 0), s.x_fov(), s.near_clip(), s.far_clip()).numpy()
        return scene_desc.camera_struct(self._mat(s._key + ".to_world"), K, s.near_clip(), s.far_clip(), w, h)


def load_scene_data(data: scenes.SceneData, device="cuda", shadows=True) -> Scene:
    return Scene(data, device=device, shadows=shadows)


def load_file(path, device="cuda", shadows=True, **_ignored) -> Scene:
    """mi.load_file(path[, parallel=False]) (examples/vocalfold_scene.py:22, main.py:29): reads the
    Mitsuba-XML subset and OBJ meshes described in fireflies_amd/loaders.py."""
    from . import loaders

    return Scene(loaders.load_mitsuba_xml(path), device=device, shadows=shadows)


def traverse(scene: Scene) -> SceneParameters:
    return scene._params


def render(scene: Scene, params: SceneParameters = None, spp: int = 16, seed: int = 0, sensor: int = 0, fp16: bool = False):
    """mi.render(scene, spp=...) -> [H,W,3] (wrapped; `.torch()` as in examples/vocalfold_scene.py:14).
    Differentiable w.r.t. `tex.data` when that parameter is a tensor that requires grad."""
    if sensor != 0:
        raise NotImplementedError("only sensor 0 renders; further sensors are projector proxies")
    p = scene._params
    tex = tex_in = None
    ch = 3
    if scene.data.projector is not None:
        tex = p["tex.data"]
        tex = tex_in = tex.t if isinstance(tex, _ArrayBase) else tex  # (tex_in: the tensor as assigned, before any conversion below)
        if not isinstance(tex, torch.Tensor):
            tex = torch.as_tensor(np.asarray(tex, np.float32))
        if tex.device != scene.device:
            tex = tex.to(scene.device)  # the reference uploads through numpy (vocalfold_scene.py:69)
        ch = 1 if tex.dim() == 2 else int(tex.shape[-1])
    scene.note_spp(spp)
    sd = scene.scene_desc(tex_channels=ch)
    if tex is None:
        tex = torch.zeros((1, 1, 1), device=scene.device)
    elif tex.dtype != torch.float32:
        tex = tex.float()
    if tex.requires_grad and torch.is_grad_enabled():
        scene.render_paths["autograd"] += 1
        img = Fn.render(tex, scene.geom, sd, scene.materials_arg(sd), spp, seed, fp16)
    else:  # nothing to differentiate: straight to the kernel (autograd.Function.apply costs ~80 us of host time per call)
        mats = scene.materials_arg(sd)
        slot = scene._render_stream(tex_in, mats) if tex_in is not None else None
        if slot is not None:  # beside the previous render, on the scene's other render stream
            rs, done, priv = slot
            scene.render_paths["two_stream"] += 1
            with torch.cuda.stream(rs):
                img = scene.geom.render_fwd(sd, mats, priv.unsqueeze(-1) if priv.dim() == 2 else priv, int(spp), int(seed), bool(fp16))
                done.record(rs)
            return _RenderedXf(img, done)
        scene.render_paths["caller_stream"] += 1
        t = tex if tex.is_contiguous() else tex.contiguous()
        img = scene.geom.render_fwd(sd, mats, t.unsqueeze(-1) if t.dim() == 2 else t, int(spp), int(seed), bool(fp16))
    return TensorXf(img)
