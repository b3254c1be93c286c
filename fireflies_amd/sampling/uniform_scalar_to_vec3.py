import torch

from . import base, torch_rng
from ..utils import math as ffmath


class UniformScalarToVec3Sampler(base.Sampler):
    """One uniform scalar broadcast to a vec3 (fireflies/sampling/uniform_scalar_to_vec3.py),
    e.g. a grey light intensity (examples/vocalfold_scene.py:80-85)."""

    def __init__(self, min, max, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        super().__init__(min, max, eval_step_size, device)

    def _vec3(self, s):
        # (the reference builds torch.tensor([s, s, s]): three .item() syncs on a device scalar; same values here)
        s = torch.as_tensor(s, dtype=torch.float32, device=self._device).reshape(-1)[0]
        return torch.stack([s, s, s])

    def sample_train(self):
        return self._vec3(ffmath.randomBetweenTensors(self._min_range, self._max_range))

    def draw(self, batch) -> int:
        if not self._train:
            return super().draw(batch)
        a, b = self._min_range, self._max_range
        assert a.size() == b.size() and a.device == b.device and a.numel() == 1
        lo, hi = self._host_bounds()
        if a.is_cuda:
            idx = batch.host.reserve(1, a.device)
            if idx is not None:
                return batch.add_uniform_reserved(idx, lo, hi, repeat=3)
        return batch.add_uniform(torch.rand(a.shape, device=a.device), lo, hi, repeat=3)

    def sample_eval(self):
        if bool((self._min_range == self._max_range).all()):
            return self._vec3(self._min_range)
        out = self._current_step  # same aliasing as Sampler.sample_eval
        self._current_step += self._eval_step_size
        if bool((self._current_step > self._max_range).any()):
            self._current_step = self._min_range
        return self._vec3(out)
