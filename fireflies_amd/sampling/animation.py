import random

import torch

from . import base


class AnimationSampler(base.Sampler):
    """Frame-index sampler for OBJ sequences (fireflies/sampling/animation.py).
    train(): python `random.randint(min_train, max_train - 1)` (:36-37).
    eval(): min_eval, min_eval+step, ... — the reference wraps with `>` (:31), so index
    `max_eval` IS emitted once per cycle (pinned by golden g8); Mesh clamps it when it selects the
    frame, because an OBJ stack of `max_eval` frames has no such index."""

    def __init__(self, min_integer_train: int, max_integer_train: int, min_integer_eval: int, max_integer_eval: int,
                 eval_step_size: int = 1, device=torch.device("cuda")) -> None:
        super().__init__(min_integer_train, max_integer_train, eval_step_size, device)
        self._min_integer_train, self._max_integer_train = min_integer_train, max_integer_train
        self._min_integer_eval, self._max_integer_eval = min_integer_eval, max_integer_eval
        self._current_step = min_integer_eval

    def sample_eval(self) -> int:
        out = self._current_step
        self._current_step += self._eval_step_size
        if self._current_step > self._max_integer_eval:
            self._current_step = self._min_integer_eval
        return out

    def sample_train(self) -> int:
        return random.randint(self._min_integer_train, self._max_integer_train - 1)

    def set_train_interval(self, min_integer_train: int, max_integer_train: int) -> None:
        self._min_integer_train, self._max_integer_train = min_integer_train, max_integer_train

    def set_eval_interval(self, min_integer_eval: int, max_integer_eval: int) -> None:
        self._min_integer_eval, self._max_integer_eval = min_integer_eval, max_integer_eval
