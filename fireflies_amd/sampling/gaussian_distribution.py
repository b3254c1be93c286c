import torch

from . import base


class GaussianSampler(base.Sampler):
    """fireflies/sampling/gaussian_distribution.py: torch.normal(mean, std); min/max are only the
    eval sweep bounds."""

    def __init__(self, min, max, mean, std, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        super().__init__(min, max, eval_step_size, device)
        self._mean, self._std = mean, std

    def sample_train(self):
        return torch.normal(self._mean, self._std)
