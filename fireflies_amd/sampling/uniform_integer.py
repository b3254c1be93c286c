import random

import torch

from . import base


class UniformIntegerSampler(base.Sampler):
    """Integers in [min_integer, max_integer) (fireflies/sampling/uniform_integer.py).
    The reference constructor forwards the BUILTINS `min`/`max` to the base class and raises
    (uniform_integer.py:17); here the integer bounds are forwarded.  train() draws
    with python's `random` (a different stream from torch's), like the reference (:29-30), but
    honours min_integer (the reference draws from [0, max))."""

    def __init__(self, min_integer: int, max_integer: int, eval_step_size: int = 1, device=torch.device("cuda")) -> None:
        super().__init__(min_integer, max_integer, eval_step_size, device)
        self._lo, self._hi = int(min_integer), int(max_integer)
        self._current_step = self._lo

    def sample_eval(self) -> int:
        out = self._current_step
        self._current_step += self._eval_step_size
        if self._current_step >= self._hi:
            self._current_step = self._lo
        return out

    def sample_train(self) -> int:
        return random.randint(self._lo, self._hi - 1)
