"""fireflies_amd — MI355X-native hot path of Henningson/Fireflies.

    import fireflies_amd as fireflies          # instead of `import fireflies`
    from fireflies_amd import mi               # instead of `import mitsuba as mi`

Layout:  csrc/ (HIP kernels + C ABI, include/ffx.h)  ->  ops / functional (tensor + autograd
plumbing)  ->  the reference's entity / sampling / projection / graphics / Scene API.
There is no CPU compute path: without libffx_hip.so or a HIP device the ops raise.
"""
from . import emitter, entity, graphics, material, projection, sampling, utils  # noqa: F401
from . import functional, mi, ops, scenes  # noqa: F401
from .scene import Scene  # noqa: F401

__version__ = "0.1.0"
