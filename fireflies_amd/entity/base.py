"""Transformable — the reference's randomisable scene entity (fireflies/entity/base.py).

Same public names, defaults and numerical behaviour, including two conventions that existing
scripts depend on (pinned by tests/golden/g7):
  * draw order per randomize(): translation, then rotation (then scale for meshes);
  * `rotate_z` feeds getPitchTransform (= rotation about Y) and `rotate_y` feeds getYawTransform
    (= rotation about Z) (entity/base.py:194-207); composition is Z @ Y @ X.
Host/device split (MI355X): the random draws happen on the entity's device with the same
torch.rand calls as the reference (so a seeded run sees the same stream).  randomize() is split in
two phases: `_draw` issues every sampler call in the reference's order and parks the drawn device
tensors in a DrawBatch; `_compose` does the 4x4 algebra on the host from the fetched values.  A whole
Scene.randomize() — or a whole batch of scene samples (Scene.randomize_batch) — therefore costs ONE
device-to-host transfer instead of one `.tolist()` sync per draw, and that transfer waits only for the
draws themselves (they run on a side stream), never for the render that is still in flight; matrices
reach the GPU as kernel arguments of ffx_scene_update_h.
"""
import math

import numpy as np
import torch

from .. import sampling
from ..utils import math as ffmath

_CPU = torch.device("cpu")
_MM4 = None


def mm4(a, b):
    """a @ b for 4x4 float32 matrices (numpy arrays or host tensors) -> numpy, through the library's ONE host definition of that product
    (include/ffx.h ffx_mat4_mul_h: an fma chain over k).  numpy's and torch's own 4x4 products differ in the last bit from CPU to CPU
    (their sgemm kernels are chosen per micro-architecture); with this the Python path of a randomisation and the native one
    (ffx_scene_randomize_h) give the same matrices everywhere."""
    global _MM4
    if _MM4 is None:
        from .._lib import api

        _MM4 = api().lib.ffx_mat4_mul_h
    a = np.ascontiguousarray(a.numpy() if isinstance(a, torch.Tensor) else a, np.float32)
    b = np.ascontiguousarray(b.numpy() if isinstance(b, torch.Tensor) else b, np.float32)
    out = np.empty((4, 4), np.float32)
    if a.shape != (4, 4) or b.shape != (4, 4) or _MM4(a.ctypes.data, b.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("mm4: two 4x4 float32 matrices expected")
    return out


class DrawBatch:
    """device tensors drawn by the samplers of one or more randomisations; `fetch()` brings all of them to
    the host with one transfer and returns them as lists of Python floats (slot -> values).

    A uniform draw `min + rand * (max - min)` (utils/math.py:170-175: three device ops in the reference) is
    registered as (rand, min, max): only `torch.rand` — the call that defines the RNG stream — runs on the
    device, the affine map is evaluated on the host after the transfer with the same float32 multiply and add,
    so the values are bit-identical."""

    def __init__(self):
        self._tensors = []
        self._slots = []  # (first tensor index, kind, repeat)
        self.host = sampling.torch_rng.HostDraws()  # draws evaluated on the host: offsets reserved per draw, values in one native call

    def add(self, t) -> int:
        self._tensors.append(t.detach().reshape(-1))
        self._slots.append((len(self._tensors) - 1, 0, 1, None, None, None))
        return len(self._slots) - 1

    def add_uniform(self, u, lo, hi, repeat: int = 1) -> int:
        """u: the torch.rand tensor (device); lo / hi: the bounds as float32 numpy arrays (host mirrors kept by the sampler)"""
        self._tensors.append(u.reshape(-1))
        self._slots.append((len(self._tensors) - 1, 1, repeat, lo, hi, None))
        return len(self._slots) - 1

    def add_uniform_host(self, u, lo, hi, repeat: int = 1) -> int:
        """u: the values of the torch.rand call as a float32 numpy array already on the host
        (sampling/torch_rng.py: the device generator's Philox stream evaluated natively); nothing to transfer"""
        self._slots.append((-1, 2, repeat, lo, hi, u))
        return len(self._slots) - 1

    def add_uniform_reserved(self, idx, lo, hi, repeat: int = 1) -> int:
        """a draw reserved in self.host (HostDraws.reserve): its values are resolved with the batch's other host draws"""
        self._slots.append((idx, 3, repeat, lo, hi, None))
        return len(self._slots) - 1

    def needs_transfer(self) -> bool:
        return any(t.is_cuda for t in self._tensors)

    def __len__(self):
        return len(self._slots)

    def fetch(self):
        return self.start_fetch().finish()

    def start_fetch(self):
        """enqueue the transfer on the current stream (CUDA entities: an asynchronous copy into pinned memory);
        `.finish()` on the returned object waits for it and returns the values"""
        if not self._slots or all(not t.is_cuda for t in self._tensors):
            return _Pending(self, None, None)
        dev = next(t.device for t in self._tensors if t.is_cuda)
        flat = torch.cat([t.to(dev, torch.float32) for t in self._tensors])
        pool = _PINNED.setdefault(flat.numel(), [])
        host = pool.pop() if pool else torch.empty(flat.shape, dtype=torch.float32).pin_memory()  # (pinning costs ~50 us: reuse)
        host.copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _Pending(self, host, ev)

    def _values(self, flat):
        if not self._slots:
            return []
        if flat is None:
            arrs = [t.to(torch.float32).numpy() for t in self._tensors]
        else:
            arrs, k = [], 0
            for t in self._tensors:
                arrs.append(flat[k : k + t.numel()])
                k += t.numel()
        out = []
        reserved = self.host.resolve() if self.host.counts else None
        for first, kind, rep, lo, hi, host_u in self._slots:
            if kind == 0:
                v = arrs[first]
            else:
                u = reserved[first] if kind == 3 else (host_u if kind == 2 else arrs[first])
                v = u * (hi - lo) + lo  # float32, one rounding per operation like the torch expression
            vals = [float(x) for x in v]
            out.append(vals * rep if rep > 1 else vals)
        return out


_PINNED = {}  # free pinned staging buffers by size


class _Pending:
    def __init__(self, batch, host, event):
        self.batch, self.host, self.event = batch, host, event

    def finish(self):
        if self.event is not None:
            self.event.synchronize()  # the one sync
            vals = self.batch._values(self.host.numpy())
            _PINNED.setdefault(self.host.numel(), []).append(self.host)
            self.host = None
            return vals
        return self.batch._values(None)


class _Lazy:
    """an attribute of an entity that a natively pushed scene sample (fireflies_amd/scene.py Scene._push_native) fills in on first use: the sample
    went to the device in one call, and what it means for the entities — matrices, last draws, attribute values — is derived only if somebody asks.
    A write counts as a use (the pending values are put in place first, so that they cannot overwrite it afterwards)."""

    def __init__(self, name):
        self.name, self.slot = name, "_lz" + name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        d = obj.__dict__
        owner = d.get("_lazy_owner")
        if owner is not None and owner._lazy is not None:
            owner._materialise()
        try:
            return d[self.slot]
        except KeyError:
            raise AttributeError(self.name) from None

    def __set__(self, obj, value):
        d = obj.__dict__
        owner = d.get("_lazy_owner")
        if owner is not None and owner._lazy is not None:
            owner._materialise()
        d[self.slot] = value


class Transformable:
    _randomized_world = _Lazy("_randomized_world")
    _last_draw = _Lazy("_last_draw")
    _host_float_attributes = _Lazy("_host_float_attributes")
    _host_vec3_attributes = _Lazy("_host_vec3_attributes")
    _randomized_float_attributes = _Lazy("_randomized_float_attributes")
    _randomized_vec3_attributes = _Lazy("_randomized_vec3_attributes")

    def __init__(self, name: str, device=torch.device("cuda")):
        self._device = device
        self._name = name
        self._randomizable = False
        self._parent = None
        self._child = None
        self._train = True
        self._float_attributes = {}
        self._randomized_float_attributes = {}
        self._vec3_attributes = {}
        self._randomized_vec3_attributes = {}
        self._host_float_attributes = {}  # host copies of the last randomised attributes (what Scene writes back)
        self._host_vec3_attributes = {}
        zeros = torch.zeros(3, device=self._device)
        self._rotation_sampler = sampling.UniformSampler(zeros.clone(), zeros.clone())
        self._translation_sampler = sampling.UniformSampler(zeros.clone(), zeros.clone())
        # host mirrors of the 4x4 state
        self._world = torch.eye(4)
        self._randomized_world = torch.eye(4)
        self._centroid_mat = torch.zeros((4, 4))
        self._eval_delta = 0.01
        self._num_updates = 0

    # ------------------------------------------------------------------ plain accessors
    def randomizable(self) -> bool:
        return self._randomizable

    def set_randomizable(self, randomizable: bool) -> None:
        sampling.base.touch()
        self._randomizable = randomizable

    def set_centroid(self, centroid) -> None:
        sampling.base.touch()
        c = centroid.detach().to(_CPU).reshape(-1)
        self._centroid_mat[0, 3], self._centroid_mat[1, 3], self._centroid_mat[2, 3] = c[0], c[1], c[2]

    def get_randomized_vec3_attributes(self) -> dict:
        if self._randomized_vec3_attributes is None:  # the tensors the reference API hands out, from the host values
            self._randomized_vec3_attributes = {k: torch.tensor(v) for k, v in self._host_vec3_attributes.items()}
        return self._randomized_vec3_attributes

    def get_randomized_float_attributes(self) -> dict:
        if self._randomized_float_attributes is None:
            self._randomized_float_attributes = {k: torch.tensor([v]) for k, v in self._host_float_attributes.items()}
        return self._randomized_float_attributes

    def vec3_attributes(self) -> dict:
        return self._vec3_attributes

    def float_attributes(self) -> dict:
        return self._float_attributes

    def parent(self):
        return self._parent

    def child(self):
        return self._child

    def name(self):
        return self._name

    # ------------------------------------------------------------------ attribute samplers
    def add_float_sampler(self, key: str, sampler) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._float_attributes[key] = sampler

    def add_float_key(self, key: str, min: float, max: float) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._float_attributes[key] = sampling.UniformSampler(min, max, device=self._device)

    def add_vec3_key(self, key: str, min, max) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._vec3_attributes[key] = sampling.UniformSampler(min, max, device=self._device)

    def add_vec3_sampler(self, key: str, sampler) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._vec3_attributes[key] = sampler

    def _all_samplers(self):
        yield self._translation_sampler
        yield self._rotation_sampler
        yield from self._float_attributes.values()
        yield from self._vec3_attributes.values()

    def train(self) -> None:
        self._train = True
        for s in self._all_samplers():
            s.train()

    def eval(self) -> None:
        self._train = False
        for s in self._all_samplers():
            s.eval()

    # ------------------------------------------------------------------ transforms
    def set_world(self, _origin) -> None:
        sampling.base.touch()
        self._world = _origin.detach().to(_CPU, torch.float32).reshape(4, 4).clone()
        self._randomized_world = self._world.clone()

    def setParent(self, parent) -> None:
        sampling.base.touch()
        self._parent = parent
        parent.setChild(self)

    def setChild(self, child) -> None:
        sampling.base.touch()
        self._child = child

    def set_rotation_sampler(self, sampler) -> None:
        sampling.base.touch()
        self._rotation_sampler = sampler

    def set_translation_sampler(self, sampler) -> None:
        sampling.base.touch()
        self._translation_sampler = sampler

    def update_index_from_sampler(self, sampler, min, max, index) -> None:
        sampler.get_min()[index] = min
        sampler.get_max()[index] = max

    def _axis_range(self, sampler, lo, hi, index):
        sampling.base.touch()
        self._randomizable = True
        self.update_index_from_sampler(sampler, lo, hi, index)

    def rotate_x(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 0)

    def rotate_y(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 1)

    def rotate_z(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 2)

    def rotate(self, min, max) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._rotation_sampler.set_sample_interval(min.to(self._device), max.to(self._device))

    def translate_x(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 0)

    def translate_y(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 1)

    def translate_z(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 2)

    def translate(self, min, max) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._translation_sampler.set_sample_interval(min.to(self._device), max.to(self._device))

    # ------------------------------------------------------------------ sampling: draw (device) / compose (host)
    @staticmethod
    def _rotation_matrix(rx, ry, rz):
        # names as in the reference: the "z" slot uses Pitch (about Y), the "y" slot Yaw (about Z); Z @ Y @ X in float32.
        # Built with numpy (same float32 products as utils.math.get*Transform + torch.matmul, bit for bit — pinned by
        # golden g7): a handful of 3x3 tensors from Python lists cost more host time than the whole refit launch.
        cz, sz = math.cos(rz), math.sin(rz)
        cy, sy = math.cos(ry), math.sin(ry)
        cx, sx = math.cos(rx), math.sin(rx)
        zMat = np.array([[cz, 0, sz, 0], [0, 1, 0, 0], [-sz, 0, cz, 0], [0, 0, 0, 1]], dtype=np.float32)  # getPitchTransform
        yMat = np.array([[cy, -sy, 0, 0], [sy, cy, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)  # getYawTransform
        xMat = np.array([[1, 0, 0, 0], [0, cx, -sx, 0], [0, sx, cx, 0], [0, 0, 0, 1]], dtype=np.float32)  # getRollTransform
        m = mm4(mm4(zMat, yMat), xMat)  # (4x4 with a zero border: the same products as the 3x3 ones, through the one product routine)
        return torch.from_numpy(m)

    @staticmethod
    def _translation_matrix(tx, ty, tz):
        m = np.eye(4, dtype=np.float32)
        m[0, 3], m[1, 3], m[2, 3] = tx, ty, tz
        return torch.from_numpy(m)

    def _draw_attributes(self, batch):
        slots = {"f": {}, "v": {}}
        for key, sampler in self._float_attributes.items():
            slots["f"][key] = sampler.draw(batch)
        for key, sampler in self._vec3_attributes.items():
            slots["v"][key] = sampler.draw(batch)
        return slots

    def _compose_attributes(self, slots, values):
        self._host_float_attributes = {k: values[i][0] for k, i in slots["f"].items()}
        self._host_vec3_attributes = {k: values[i] for k, i in slots["v"].items()}
        self._randomized_float_attributes = self._randomized_vec3_attributes = None  # rebuilt on demand (get_randomized_*)

    def _draw(self, batch):
        """phase 1 of randomize(): the sampler calls of the reference, in its order (translation, rotation,
        float / vec3 attributes; entity/base.py:220-234).  Returns a ticket for _compose, None if not randomisable."""
        if not self.randomizable():
            return None
        return {"t": self._translation_sampler.draw(batch), "r": self._rotation_sampler.draw(batch), "a": self._draw_attributes(batch)}

    def _compose(self, ticket, values) -> None:
        """phase 2: (T + centroid) @ R @ world on the host from the fetched draws"""
        if ticket is None:
            return
        t = self._translation_matrix(*values[ticket["t"]])
        self._last_translation = t
        self._last_draw = (values[ticket["t"]], values[ticket["r"]])
        rot = self._rotation_matrix(*values[ticket["r"]])
        self._randomized_world = torch.from_numpy(mm4(mm4(t.numpy() + self._centroid_mat.numpy(), rot), self._world))
        self._compose_attributes(ticket["a"], values)

    # the last drawn translation / rotation (attributes of the reference), as tensors on demand
    @property
    def _random_translation(self):
        return torch.tensor(self._last_draw[0]) if getattr(self, "_last_draw", None) else None

    @_random_translation.setter
    def _random_translation(self, v):
        self._last_draw = ([float(x) for x in v.reshape(-1).tolist()], (getattr(self, "_last_draw", None) or (None, [0.0, 0.0, 0.0]))[1])

    @property
    def _sampled_rotation(self):
        return torch.tensor(self._last_draw[1]) if getattr(self, "_last_draw", None) else None

    @_sampled_rotation.setter
    def _sampled_rotation(self, v):
        self._last_draw = ((getattr(self, "_last_draw", None) or ([0.0, 0.0, 0.0], None))[0], [float(x) for x in v.reshape(-1).tolist()])

    def sample_rotation(self):
        self._sampled_rotation = self._rotation_sampler.sample()
        return self._rotation_matrix(*(float(v) for v in self._sampled_rotation.reshape(-1).tolist())).to(self._device)

    def sample_translation(self):
        self._random_translation = self._translation_sampler.sample()
        self._last_translation = self._translation_matrix(*(float(v) for v in self._random_translation.reshape(-1).tolist()))
        return self._last_translation.to(self._device)

    def randomize(self) -> None:
        batch = DrawBatch()
        ticket = self._draw(batch)
        self._compose(ticket, batch.fetch())

    def relative(self) -> bool:
        return self._parent is not None

    def _world_host(self):
        if self._parent is None:
            return self._randomized_world.clone()
        return torch.from_numpy(mm4(self._parent._world_host(), self._randomized_world))

    def world(self):
        return self._world_host().to(self._device)

    def nonRandomizedWorld(self):
        if self._parent is None:
            return self._world.to(self._device)
        return (self._parent.nonRandomizedWorld().to(_CPU) @ self._world).to(self._device)

    def origin(self):
        """world-space position (used through Camera.origin, projection/camera.py:58-59)."""
        return self.world()[0:3, 3]
