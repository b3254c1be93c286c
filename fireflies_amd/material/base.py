import warnings

import torch

from ..entity import Transformable


def _no_transform(kind):
    """Materials have no pose: transform setters warn and do nothing else harmful.  (In the
    reference two of the four warning decorators forget to return the wrapper, so
    Material.translate_*, set_world and world are `None` attributes — fireflies/utils/
    warnings.py:39-66; here they are callable and warn.)"""

    def deco(fn):
        def wrapper(self, *a, **k):
            warnings.warn(f"{kind} assignment has no effect on a Material ({fn.__name__})", stacklevel=2)
            return fn(self, *a, **k)

        wrapper.__name__ = fn.__name__
        return wrapper

    return deco


class Material(Transformable):
    """fireflies/material/base.py: randomize() only samples the float / vec3 attributes
    (material/base.py:22-27)."""

    def __init__(self, name: str, device=torch.device("cuda")):
        super().__init__(name, device)

    def _draw(self, batch):
        return {"a": self._draw_attributes(batch)}

    def _compose(self, ticket, values) -> None:
        self._compose_attributes(ticket["a"], values)


for _n, _k in [("set_world", "World"), ("setParent", "Relative"), ("setChild", "Relative"), ("rotate_x", "Rotation"), ("rotate_y", "Rotation"),
               ("rotate_z", "Rotation"), ("rotate", "Rotation"), ("translate_x", "Translation"), ("translate_y", "Translation"),
               ("translate_z", "Translation"), ("translate", "Translation")]:
    setattr(Material, _n, _no_transform(_k)(getattr(Transformable, _n)))
