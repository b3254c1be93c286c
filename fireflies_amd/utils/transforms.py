"""The reference ships this module EMPTY (fireflies/utils/transforms.py, 0 bytes) although
projection/camera.py:68-73 and projection/laser.py:59,85,213,307-318 call
`fireflies.utils.transforms.transform_points`.  Here it re-exports utils.math so those
call sites work."""
from .math import (  # noqa: F401
    convert_points_from_homogeneous, convert_points_to_homogeneous, convert_points_to_nonhomogeneous, getPitchTransform, getRollTransform,
    getXTransform, getYawTransform, getYTransform, getZTransform, normalize, randomBetweenTensors,
)
from .math import toMat4x4, transform_directions, transform_points  # noqa: F401
