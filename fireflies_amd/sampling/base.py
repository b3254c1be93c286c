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

# bumped by every mutation of a sampler / entity configuration that goes through the API: Scene's pre-drawn
# randomisation (scene.py: _predraw) is only used if nothing changed since it was drawn
_MUTATIONS = [0]


def touch() -> None:
    _MUTATIONS[0] += 1


def mutation_count() -> int:
    return _MUTATIONS[0]


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
        touch()
        self._min_range, self._max_range = min.clone(), max.clone()

    def get_min(self):
        touch()  # hands out the live tensor: the caller may write through it
        return self._min_range

    def get_max(self):
        touch()
        return self._max_range

    def set_sample_max(self, max) -> None:
        touch()
        self._max_range = max.clone()

    def set_sample_min(self, min) -> None:
        touch()
        self._min_range = min.clone()

    def train(self) -> None:
        touch()
        self._train = True

    def eval(self) -> None:
        touch()
        self._train = False

    def sample(self):
        return self.sample_train() if self._train else self.sample_eval()

    def sample_train(self):
        raise NotImplementedError

    def _host_bounds(self):
        """float32 numpy mirrors of (min, max), refreshed when the configuration counter moved (one sync then,
        for device bounds)"""
        hb = getattr(self, "_hb", None)
        # (the tensors' version counters catch in-place writes through a tensor that get_min() / get_max() handed out EARLIER — the
        # accessor itself bumps the configuration counter, a later `t += 1` on its result does not)
        key = (_MUTATIONS[0], self._min_range._version, self._max_range._version, id(self._min_range), id(self._max_range))
        if hb is None or hb[0] != key:
            lo = self._min_range.detach().reshape(-1).to("cpu", torch.float32).numpy().copy()
            hi = self._max_range.detach().reshape(-1).to("cpu", torch.float32).numpy().copy()
            hb = self._hb = (key, lo, hi)
        return hb[1], hb[2]

    def draw(self, batch) -> int:
        """registers this sampler's next draw in an entity.DrawBatch and returns its slot (the value reaches
        the host with the batch's single transfer).  Subclasses whose train draw is min + rand * (max - min)
        register only the rand (see DrawBatch.add_uniform)."""
        return batch.add(torch.as_tensor(self.sample(), dtype=torch.float32))

    def sample_eval(self):
        if bool((self._min_range == self._max_range).all()):
            return self._min_range
        out = self._current_step  # shares storage on purpose, see module docstring
        self._current_step += self._eval_step_size
        if bool((self._current_step > self._max_range).any()):
            self._current_step = self._min_range
        return out
