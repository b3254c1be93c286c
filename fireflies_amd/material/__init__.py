from .base import Material  # noqa: F401
