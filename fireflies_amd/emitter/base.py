import torch

from ..entity import Transformable


class Light(Transformable):
    """fireflies/emitter/base.py: a Transformable whose float / vec3 attributes (e.g.
    `intensity.value`) are written back into the scene parameters by Scene.update_lights."""

    def __init__(self, name: str, device=torch.device("cuda")):
        super().__init__(name, device)
