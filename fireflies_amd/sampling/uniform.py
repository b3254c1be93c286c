import torch

from . import base, torch_rng
from ..utils import math as ffmath


class UniformSampler(base.Sampler):
    """fireflies/sampling/uniform.py: train draw = min + rand * (max - min), ONE torch.rand of the
    bound's shape on the bound's device (utils/math.py:170-175) — this fixes the RNG stream."""

    def __init__(self, min, max, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        super().__init__(min, max, eval_step_size, device)

    def sample_train(self):
        return ffmath.randomBetweenTensors(self._min_range, self._max_range)

    def draw(self, batch) -> int:
        if not self._train:
            return super().draw(batch)
        a, b = self._min_range, self._max_range
        assert a.size() == b.size() and a.device == b.device
        lo, hi = self._host_bounds()
        if a.is_cuda:
            idx = batch.host.reserve(a.numel(), a.device)  # that torch.rand call's place in the generator's stream, without the launch
            if idx is not None:
                return batch.add_uniform_reserved(idx, lo, hi)
        return batch.add_uniform(torch.rand(a.shape, device=a.device), lo, hi)  # the same torch.rand call as sample_train
