from .base import Light  # noqa: F401
