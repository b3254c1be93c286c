from .base import Transformable
from .mesh import Mesh

__all__ = ["Transformable", "Mesh"]
