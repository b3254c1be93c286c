"""Camera — fireflies/projection/camera.py: a pose (Transformable) plus the 4x4
`mi.perspective_projection` matrix, field of view and clip planes."""
import torch

from ..utils import transforms


class Camera:
    id = 0
    MITSUBA_KEYS = {"fov": "x_fov", "f": "x_fov", "to_world": "to_world", "world": "to_world"}

    def __init__(self, transform, perspective, fov: float, near_clip: float = 0.01, far_clip: float = 1000.0, device=torch.device("cuda")):
        self.device = device
        self._transformable = transform
        self._perspective = perspective
        self._near_clip = near_clip
        self._far_clip = far_clip
        self._fov = fov
        self._key = self.generate_mitsuba_key()
        Camera.id += 1

    def generate_mitsuba_key(self) -> str:
        # the reference formats the *builtin* `id` here (camera.py:46-50); the counter is meant
        return "PerspectiveCamera" if Camera.id == 0 else "PerspectiveCamera_{0}".format(Camera.id)

    def full_key(self, key: str):
        return self._key + "." + Camera.MITSUBA_KEYS[key]

    def key(self) -> str:
        return self._key

    def near_clip(self) -> float:
        return self._near_clip

    def far_clip(self) -> float:
        return self._far_clip

    def fov(self):
        return self._fov

    def origin(self):
        return self._transformable.origin()

    def world(self):
        return self._transformable.world()

    def randomize(self) -> None:
        self._transformable.randomize()

    def pointsToNDC(self, points):
        """world points -> camera space -> sample space (camera.py:67-74; works here because
        utils.transforms is not empty)."""
        view = transforms.transform_points(points, self.world().inverse().to(points.device))
        return transforms.transform_points(view, self._perspective.to(points.device))
