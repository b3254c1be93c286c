from . import rasterization, depth  # noqa: F401
