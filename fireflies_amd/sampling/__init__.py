from .base import Sampler
from .gaussian_distribution import GaussianSampler
from .uniform import UniformSampler
from .uniform_integer import UniformIntegerSampler
from .uniform_scalar_to_vec3 import UniformScalarToVec3Sampler
from .animation import AnimationSampler
from .noise_texture_lerp import NoiseTextureLerpSampler

__all__ = ["Sampler", "GaussianSampler", "UniformSampler", "UniformIntegerSampler", "UniformScalarToVec3Sampler", "AnimationSampler", "NoiseTextureLerpSampler"]
