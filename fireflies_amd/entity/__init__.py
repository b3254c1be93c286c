from .base import DrawBatch, Transformable
from .mesh import Mesh

__all__ = ["Transformable", "Mesh", "DrawBatch"]
