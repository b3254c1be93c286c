"""Transformable — the reference's randomisable scene entity (fireflies/entity/base.py).

Same public names, defaults and numerical behaviour, including two conventions that existing
scripts depend on (pinned by tests/golden/g7):
  * draw order per randomize(): translation, then rotation (then scale for meshes);
  * `rotate_z` feeds getPitchTransform (= rotation about Y) and `rotate_y` feeds getYawTransform
    (= rotation about Z) (entity/base.py:194-207); composition is Z @ Y @ X.
Host/device split (MI355X): the random draws happen on the entity's device with the same
torch.rand calls as the reference (so a seeded run sees the same stream), but the 4x4 algebra
is done on the host from ONE `.tolist()` per draw instead of dozens of one-element device
kernels and per-angle syncs; matrices are handed to the GPU once per randomisation as a single
[S,16] buffer (Scene.update_meshes -> ffx_scene_update).
"""
import torch

from .. import sampling
from ..utils import math as ffmath

_CPU = torch.device("cpu")


class Transformable:
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
        self._randomizable = randomizable

    def set_centroid(self, centroid) -> None:
        c = centroid.detach().to(_CPU).reshape(-1)
        self._centroid_mat[0, 3], self._centroid_mat[1, 3], self._centroid_mat[2, 3] = c[0], c[1], c[2]

    def get_randomized_vec3_attributes(self) -> dict:
        return self._randomized_vec3_attributes

    def get_randomized_float_attributes(self) -> dict:
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
        self._randomizable = True
        self._float_attributes[key] = sampler

    def add_float_key(self, key: str, min: float, max: float) -> None:
        self._randomizable = True
        self._float_attributes[key] = sampling.UniformSampler(min, max, device=self._device)

    def add_vec3_key(self, key: str, min, max) -> None:
        self._randomizable = True
        self._vec3_attributes[key] = sampling.UniformSampler(min, max, device=self._device)

    def add_vec3_sampler(self, key: str, sampler) -> None:
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
        self._world = _origin.detach().to(_CPU, torch.float32).reshape(4, 4).clone()
        self._randomized_world = self._world.clone()

    def setParent(self, parent) -> None:
        self._parent = parent
        parent.setChild(self)

    def setChild(self, child) -> None:
        self._child = child

    def set_rotation_sampler(self, sampler) -> None:
        self._rotation_sampler = sampler

    def set_translation_sampler(self, sampler) -> None:
        self._translation_sampler = sampler

    def update_index_from_sampler(self, sampler, min, max, index) -> None:
        sampler.get_min()[index] = min
        sampler.get_max()[index] = max

    def _axis_range(self, sampler, lo, hi, index):
        self._randomizable = True
        self.update_index_from_sampler(sampler, lo, hi, index)

    def rotate_x(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 0)

    def rotate_y(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 1)

    def rotate_z(self, min_rot: float, max_rot: float) -> None:
        self._axis_range(self._rotation_sampler, min_rot, max_rot, 2)

    def rotate(self, min, max) -> None:
        self._randomizable = True
        self._rotation_sampler.set_sample_interval(min.to(self._device), max.to(self._device))

    def translate_x(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 0)

    def translate_y(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 1)

    def translate_z(self, min_translation: float, max_translation: float) -> None:
        self._axis_range(self._translation_sampler, min_translation, max_translation, 2)

    def translate(self, min, max) -> None:
        self._randomizable = True
        self._translation_sampler.set_sample_interval(min.to(self._device), max.to(self._device))

    # ------------------------------------------------------------------ sampling (host 4x4 algebra)
    def _sample_rotation_host(self):
        self._sampled_rotation = self._rotation_sampler.sample()
        rx, ry, rz = (float(v) for v in self._sampled_rotation.reshape(-1).tolist())
        # names as in the reference: the "z" slot uses Pitch (about Y), the "y" slot Yaw (about Z)
        zMat = ffmath.getPitchTransform(rz, _CPU)
        yMat = ffmath.getYawTransform(ry, _CPU)
        xMat = ffmath.getRollTransform(rx, _CPU)
        return ffmath.toMat4x4(zMat @ yMat @ xMat)

    def _sample_translation_host(self):
        self._random_translation = self._translation_sampler.sample()
        tx, ty, tz = (float(v) for v in self._random_translation.reshape(-1).tolist())
        t = torch.eye(4)
        t[0, 3], t[1, 3], t[2, 3] = tx, ty, tz
        self._last_translation = t
        return t

    def sample_rotation(self):
        return self._sample_rotation_host().to(self._device)

    def sample_translation(self):
        return self._sample_translation_host().to(self._device)

    def _sample_attributes(self):
        for key, sampler in self._float_attributes.items():
            self._randomized_float_attributes[key] = sampler.sample()
        for key, sampler in self._vec3_attributes.items():
            self._randomized_vec3_attributes[key] = sampler.sample()

    def randomize(self) -> None:
        if not self.randomizable():
            return
        self._randomized_world = (self._sample_translation_host() + self._centroid_mat) @ self._sample_rotation_host() @ self._world
        self._sample_attributes()

    def relative(self) -> bool:
        return self._parent is not None

    def _world_host(self):
        if self._parent is None:
            return self._randomized_world.clone()
        return self._parent._world_host() @ self._randomized_world

    def world(self):
        return self._world_host().to(self._device)

    def nonRandomizedWorld(self):
        if self._parent is None:
            return self._world.to(self._device)
        return (self._parent.nonRandomizedWorld().to(_CPU) @ self._world).to(self._device)

    def origin(self):
        """world-space position (used through Camera.origin, projection/camera.py:58-59)."""
        return self.world()[0:3, 3]
