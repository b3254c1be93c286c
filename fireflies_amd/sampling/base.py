"""Sampler base — behaviour of fireflies/sampling/base.py.

train(): random draws (sample_train of the subclass); eval(): deterministic sweep from min to
max in `eval_step_size` increments (base.py:64-74).  The reference's sweep has two aliasing
effects that change the numbers it returns (pinned by tests/golden/g8):
  * `sample = self._current_step` followed by an in-place `+=` returns the INCREMENTED value;
  * on wrap-around `self._current_step = self._min_range` shares storage with the lower bound,
    which then drifts upward with every later step.
They are reproduced (same statements on shared tensors) because existing scripts observe them.
"""
import torch


class Sampler:
    def __init__(self, min, max, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        self._device = device
        as_tensor = lambda v: v.clone() if type(v) is torch.Tensor else torch.tensor([v], device=device)  # noqa: E731
        self._min_range = as_tensor(min)
        self._max_range = as_tensor(max)
        self._current_step = as_tensor(min)
        self._eval_step_size = eval_step_size
        self._train = True

    # interval accessors (base.py:32-46); get_min/get_max hand out the live tensors, which the
    # Transformable convenience setters rely on (entity/base.py:146-151)
    def set_sample_interval(self, min, max) -> None:
        self._min_range, self._max_range = min.clone(), max.clone()

    def get_min(self):
        return self._min_range

    def get_max(self):
        return self._max_range

    def set_sample_max(self, max) -> None:
        self._max_range = max.clone()

    def set_sample_min(self, min) -> None:
        self._min_range = min.clone()

    def train(self) -> None:
        self._train = True

    def eval(self) -> None:
        self._train = False

    def sample(self):
        return self.sample_train() if self._train else self.sample_eval()

    def sample_train(self):
        raise NotImplementedError

    def sample_eval(self):
        if bool((self._min_range == self._max_range).all()):
            return self._min_range
        out = self._current_step  # shares storage on purpose, see module docstring
        self._current_step += self._eval_step_size
        if bool((self._current_step > self._max_range).any()):
            self._current_step = self._min_range
        return out
