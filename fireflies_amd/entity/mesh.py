"""Mesh — fireflies/entity/mesh.py: a Transformable with a scale sampler and vertex animation.

randomize() composes (T + centroid) @ R @ S @ world with draws in the order translation,
rotation, scale (mesh.py:141-150; float/vec3 attributes are NOT sampled for meshes, like the
reference).  Vertex animation is either a python function of (vertices, t) (mesh.py:66-72) or a
stack of frames with identical topology loaded from OBJ files (mesh.py:74-109,167-181); the frame
is picked when the scene pushes the update, i.e. after every entity has been randomised.

MI355X note: `get_randomized_vertices()` keeps the reference semantics (a torch expression) for
API users, but Scene.update_meshes never calls it — it hands (frame, world matrix) to
ffx_scene_update, which transforms the vertices inside the BVH-refit pass.
"""
import os

import numpy as np
import torch

from . import base
from .. import sampling
from ..utils import math as ffmath


def load_obj_vertices(path: str) -> torch.Tensor:
    """All `v x y z` records of a Wavefront OBJ file, in file order ([V,3] float32).
    Stands in for pywavefront.Wavefront(...).vertices (mesh.py:173-179; pywavefront is not a
    dependency here)."""
    from ..loaders import load_obj

    return torch.from_numpy(load_obj(path)[0])


class Mesh(base.Transformable):
    def __init__(self, name: str, vertex_data, device=torch.device("cuda")):
        super().__init__(name, device)
        self._vertices = vertex_data.to(self._device)
        self._vertices_animation = None
        ones = torch.ones(3, device=self._device)
        self._scale_sampler = sampling.UniformSampler(ones.clone(), ones.clone())
        self._animated = False
        self._anim_data_train = None
        self._anim_data_eval = None
        self._animation_func = None
        self._animation_sampler = None
        self._last_time_sample = None
        self._faces = None

    # ------------------------------------------------------------------ scale
    def set_scale_sampler(self, sampler) -> None:
        sampling.base.touch()
        self._scale_sampler = sampler

    def scale_x(self, min_scale: float, max_scale: float) -> None:
        self._axis_range(self._scale_sampler, min_scale, max_scale, 0)

    def scale_y(self, min_scale: float, max_scale: float) -> None:
        self._axis_range(self._scale_sampler, min_scale, max_scale, 1)

    def scale_z(self, min_scale: float, max_scale: float) -> None:
        self._axis_range(self._scale_sampler, min_scale, max_scale, 2)

    def scale(self, min, max) -> None:
        sampling.base.touch()
        self._randomizable = True
        self._scale_sampler.set_sample_interval(min.to(self._device), max.to(self._device))

    # ------------------------------------------------------------------ animation
    def animated(self) -> bool:
        return self._animated

    def add_animation(self, animation_data) -> None:
        """frames [T,V,3] used for both train and eval (the reference stores them in an attribute
        nothing reads, mesh.py:61-64; here they become the frame stack)."""
        data = animation_data.to(self._device)
        self._animation_vertices = data
        self.set_animation_frames(data, data)

    def set_animation_frames(self, train_frames, eval_frames=None) -> None:
        """Frame stacks without going through OBJ files (same state as add_*_animation_from_obj)."""
        sampling.base.touch()
        self._anim_data_train = train_frames.to(self._device)
        self._anim_data_eval = (train_frames if eval_frames is None else eval_frames).to(self._device)
        if self._animation_sampler is None:
            self._animation_sampler = sampling.AnimationSampler(0, 1, 0, 1, device=self._device)
        self._animation_sampler.set_train_interval(0, self._anim_data_train.shape[0])
        self._animation_sampler.set_eval_interval(0, self._anim_data_eval.shape[0])
        self._animated = True
        self._randomizable = True

    def set_pool_animation(self, n_train: int, n_eval: int, train_start: int = 0, eval_start: int = None) -> None:
        """Animation frames that already live in the scene's device vertex pool (procedural scenes,
        fireflies_amd.scenes): frames [train_start, train_start + n_train) are the train set,
        [eval_start, eval_start + n_eval) the eval set.  Selecting a frame is then a pointer change
        inside ffx_scene_update — no vertex data moves."""
        sampling.base.touch()
        eval_start = train_start + n_train if eval_start is None else eval_start
        self._pool_frames = {"train": (int(train_start), int(n_train)), "eval": (int(eval_start), int(n_eval))}
        self._animation_sampler = sampling.AnimationSampler(0, int(n_train), 0, int(n_eval), device=self._device)
        self._animated = True
        self._randomizable = True

    def add_animation_func(self, func, min_range, max_range) -> None:
        sampling.base.touch()
        self._animation_func = func
        self._animation_sampler = sampling.UniformSampler(min_range, max_range, device=self._device)
        self._animated = True
        self._randomizable = True

    def _ensure_anim_sampler(self):
        if self._animation_sampler is None:
            self._animation_sampler = sampling.AnimationSampler(0, 1, 0, 1, device=self._device)

    def add_train_animation_from_obj(self, path: str, min: int = None, max: int = None) -> None:
        sampling.base.touch()
        self._anim_data_train = self.load_animation(path)
        self._ensure_anim_sampler()
        # the reference ignores `min` (mesh.py:81,88): the interval always starts at 0
        self._animation_sampler.set_train_interval(0, self._anim_data_train.shape[0] if max is None else max)
        self._animated = True

    def add_eval_animation_from_obj(self, path: str, min: int = None, max: int = None) -> None:
        sampling.base.touch()
        self._anim_data_eval = self.load_animation(path)
        self._ensure_anim_sampler()
        self._animation_sampler.set_eval_interval(0, self._anim_data_eval.shape[0] if max is None else max)
        self._animated = True  # the reference only sets this on the train call (mesh.py:92); harmless

    def train(self) -> None:
        super().train()
        self._scale_sampler.train()
        if self._animation_sampler:
            self._animation_sampler.train()

    def eval(self) -> None:
        super().eval()
        self._scale_sampler.eval()
        if self._animation_sampler:
            self._animation_sampler.eval()

    def set_faces(self, faces) -> None:
        self._faces = faces.to(self._device)

    def set_vertices(self, vertices) -> None:
        sampling.base.touch()
        self._vertices = vertices.to(self._device)

    def faces(self):
        return self._faces

    def get_vertices(self):
        return self._vertices

    # ------------------------------------------------------------------ randomisation
    def sample_scale(self):
        sx, sy, sz = (float(v) for v in self._scale_sampler.sample().reshape(-1).tolist())
        m = torch.eye(4)
        m[0, 0], m[1, 1], m[2, 2] = sx, sy, sz
        return m.to(self._device)

    def _draw(self, batch):
        """translation, rotation, scale (mesh.py:141-150; float / vec3 attributes are NOT sampled for meshes)"""
        if not self.randomizable():
            return None
        return {"t": self._translation_sampler.draw(batch), "r": self._rotation_sampler.draw(batch), "s": self._scale_sampler.draw(batch)}

    def _compose(self, ticket, values) -> None:
        if ticket is None:
            return
        t = self._translation_matrix(*values[ticket["t"]])
        self._last_translation = t
        self._last_draw = (values[ticket["t"]], values[ticket["r"]])
        sx, sy, sz = values[ticket["s"]]
        sc = np.zeros((4, 4), dtype=np.float32)
        sc[0, 0], sc[1, 1], sc[2, 2], sc[3, 3] = sx, sy, sz, 1.0
        rot = self._rotation_matrix(*values[ticket["r"]])
        # (numpy: the same float32 products as the torch expression, see Transformable._rotation_matrix)
        self._randomized_world = torch.from_numpy(base.mm4(base.mm4(base.mm4(t.numpy() + self._centroid_mat.numpy(), rot), sc), self._world))

    def load_animation(self, path: str):
        frames = [load_obj_vertices(os.path.join(path, f)) for f in sorted(os.listdir(path)) if f.endswith(".obj")]
        if not frames:
            raise FileNotFoundError(f"no .obj files in {path}")
        return torch.stack(frames).to(self._device)

    def sample_animation_index(self):
        """Draws the animation time/frame (advances the sampler) and returns
        ("frames", "train"|"eval", index) or ("func", None, t) or None."""
        if not self._animated:
            return None
        t = self._animation_sampler.sample()
        self._last_time_sample = t
        if self._animation_func is not None:
            return ("func", None, t)
        pool = getattr(self, "_pool_frames", None)
        if pool is not None:
            which = "train" if self._train else "eval"
            return ("frames", which, int(min(max(int(t), 0), pool[which][1] - 1)))
        if self._anim_data_train is not None and self._anim_data_eval is not None:
            which = "train" if self._train else "eval"
            stack = self._anim_data_train if self._train else self._anim_data_eval
            return ("frames", which, int(min(max(int(t), 0), stack.shape[0] - 1)))
        return None

    def sample_animation(self):
        # (inside Scene.randomize the frame was already drawn with the other samplers — scene.Scene._draw_all — and is
        # waiting in _pending_pick: drawing again would consume a second animation draw per randomisation)
        pick = self._pending_pick if hasattr(self, "_pending_pick") else self.sample_animation_index()
        if pick is None:
            return self._vertices if not self._animated else None
        if pick[0] == "func":
            return self._animation_func(self._vertices, pick[2])
        return (self._anim_data_train if pick[1] == "train" else self._anim_data_eval)[pick[2]]

    def get_randomized_vertices(self):
        v = self.sample_animation() if self._animated else self._vertices
        return ffmath.transform_points(v, self.world())
