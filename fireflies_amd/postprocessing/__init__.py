"""Image post-processing of the dataset path — same classes as
fireflies/postprocessing/*: a probability-gated chain of blur / silhouette / noise.

The reference runs these on the HOST with numpy, kornia and cv2 on images that were copied back
from the GPU (main.py:138-160).  Here an image that is a HIP tensor stays on the device (the blur
is the K3 kernel); a numpy image is accepted too and a numpy image is returned, like the reference.
The python-`random` draws happen in the same order (gate, then the function's own draws).
"""
import random

import numpy as np
import torch

from .. import ops


def _to_device(image):
    if isinstance(image, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).cuda(), True
    return image.float().contiguous(), False


def _back(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


class BasePostProcessingFunction:
    def __init__(self, probability: float):
        self._probability = probability

    def apply(self, image):
        """base.py:10-14: applied with probability `probability` (one random.uniform draw)."""
        if random.uniform(0, 1) < self._probability:
            return self.post_process(image)
        return image

    def post_process(self, image):
        raise NotImplementedError


class PostProcessor:
    def __init__(self, post_process_funcs):
        self._post_process_functs = post_process_funcs

    def post_process(self, image):
        out = image.copy() if isinstance(image, np.ndarray) else image.clone()
        for func in self._post_process_functs:
            out = func.apply(out)
        return out


class GaussianBlur(BasePostProcessingFunction):
    """gauss_blur.py:7-28: kornia.filters.gaussian_blur2d(image, kernel_size, sigma) on a [H,W] image."""

    def __init__(self, kernel_size, sigma, probability: float):
        super().__init__(probability)
        if kernel_size[0] != kernel_size[1] or sigma[0] != sigma[1]:
            raise NotImplementedError("only square kernels (all call sites of the reference use them)")
        self._kernel_size, self._sigma = kernel_size, sigma

    def post_process(self, image):
        t, was_np = _to_device(image)
        return _back(ops.blur_fwd(t, int(self._kernel_size[0]), float(self._sigma[0])), was_np)


class WhiteNoise(BasePostProcessingFunction):
    """white_noise.py:5-20: image + N(mean, std), clipped to [0,1].  rng="numpy" (default) draws the noise exactly as
    the reference does — `np.random.normal(ones * mean, ones * std)` from the GLOBAL numpy generator, float64 — so
    a script seeded with np.random.seed sees the same images (pinned by tests/golden/g12); the draw is host work and
    one upload per image.  rng="device" draws with torch.randn_like on the image's device instead (no host work,
    torch's generator: a different stream)."""

    def __init__(self, mean: float, std: float, probability: float, rng: str = "numpy"):
        super().__init__(probability)
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self._mean, self._std, self._rng = mean, std, rng

    def post_process(self, image):
        t, was_np = _to_device(image)
        if self._rng == "numpy":
            shape = tuple(t.shape)
            noise = np.random.normal(np.ones(shape) * self._mean, np.ones(shape) * self._std)  # float64, the reference's call
            out = (t.double() + torch.from_numpy(noise).to(t.device)).to(t.dtype)  # `image += noise`: one rounding to the image type
        else:  # (one launch behind the draw: ffx_noise_clamp — t + (n std + mean), clipped, every operation rounded as the torch expression rounds it)
            return _back(ops.noise_clamp(t, torch.randn_like(t), self._mean, self._std, 0.0, 1.0), was_np)
        return _back(torch.clamp(out, 0, 1), was_np)


class ApplySilhouette(BasePostProcessingFunction):
    """apply_silhouette.py:10-40: multiply by a blurred filled circle (endoscope vignette); centre and
    radius drawn with random.randint in the reference's order (cc_x, cc_y, radius)."""

    def __init__(self, probability: float = 2.0):
        super().__init__(probability)

    def post_process(self, image):
        t, was_np = _to_device(image)
        cc_x = random.randint(100, 200)
        cc_y = random.randint(200, 300)
        radius = random.randint(170, 230)
        # the filled circle, its 11 x 11 sigma-5 blur and the product in ONE launch (ffx_silhouette_fwd: ops.blur_fwd(mask) * t bit for bit; as torch
        # expressions — two coordinate grids, the disc test, the blur, the product — it was nine launches of the dataset loop's ~25 per sample)
        return _back(ops.silhouette(t, cc_x, cc_y, radius, 11, 5.0), was_np)


def rgb_to_gray(image):
    """main.py:157 `cv2.cvtColor(render, cv2.COLOR_RGB2GRAY)` for an image that stays on the device: [H,W,3] -> [H,W], (r 0.299 + g 0.587) + b 0.114
    in float32 (one launch, ffx_rgb_to_gray); a numpy image is converted on the host with the same expression."""
    if isinstance(image, np.ndarray):
        im = image.astype(np.float32, copy=False)
        return (im[..., 0] * np.float32(0.299) + im[..., 1] * np.float32(0.587)) + im[..., 2] * np.float32(0.114)
    return ops.rgb_to_gray(image.contiguous())


__all__ = ["BasePostProcessingFunction", "PostProcessor", "GaussianBlur", "WhiteNoise", "ApplySilhouette", "rgb_to_gray"]
