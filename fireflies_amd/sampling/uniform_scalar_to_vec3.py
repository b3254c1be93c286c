import torch

from . import base
from ..utils import math as ffmath


class UniformScalarToVec3Sampler(base.Sampler):
    """One uniform scalar broadcast to a vec3 (fireflies/sampling/uniform_scalar_to_vec3.py),
    e.g. a grey light intensity (examples/vocalfold_scene.py:80-85)."""

    def __init__(self, min, max, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        super().__init__(min, max, eval_step_size, device)

    def _vec3(self, s):
        return torch.tensor([s, s, s], device=self._device)

    def sample_train(self):
        return self._vec3(ffmath.randomBetweenTensors(self._min_range, self._max_range))

    def sample_eval(self):
        if bool((self._min_range == self._max_range).all()):
            return self._vec3(self._min_range)
        out = self._current_step  # same aliasing as Sampler.sample_eval
        self._current_step += self._eval_step_size
        if bool((self._current_step > self._max_range).any()):
            self._current_step = self._min_range
        return self._vec3(out)
