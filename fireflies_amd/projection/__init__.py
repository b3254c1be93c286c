from .camera import Camera
from .laser import Laser

__all__ = ["Camera", "Laser"]
