"""Perlin-noise base-colour textures — behaviour of fireflies/sampling/noise_texture_lerp.py.

The reference's dataset loop draws one of these per iteration and assigns it to the mucosa's texture-valued base colour
(`mat-Mucosa.brdf_0.base_color.data`, main.py:138-153).  Same draws from the same generators in the same order, so a seeded
script sees the same textures (pinned by tests/golden/g13, captured from the reference): per sample
`random.randint(1, 6)` (lattice 2^i), `random.randint(1, 4)` (octaves), `random.uniform(0.1, 2)` (persistence), then one
`torch.rand(r + 1, r + 1)` of lattice angles per octave from torch's CPU generator.  The lattice (a few hundred KB at most)
is drawn on the host as in the reference; the per-texel arithmetic runs on the sampler's device.
"""
import math
import random

import torch

from . import base


def _fade(t):
    return 6 * t**5 - 15 * t**4 + 10 * t**3


def perlin_2d(shape, res, device="cpu"):
    """one octave of 2-D gradient noise, [shape0, shape1], lattice res0 x res1 cells (rand_perlin_2d, :8-47)"""
    h, w = int(shape[0]), int(shape[1])
    r0, r1 = int(res[0]), int(res[1])
    cell = (h // r0, w // r1)
    if cell[0] < 1 or cell[1] < 1:
        raise ValueError(f"a {r0} x {r1} lattice needs at least as many texels per axis (got {h} x {w})")
    angles = 2 * math.pi * torch.rand(r0 + 1, r1 + 1)  # torch's CPU generator, as in the reference
    grad = torch.stack((torch.cos(angles), torch.sin(angles)), dim=-1).to(device)
    # position of every texel inside its lattice cell, and the cell it is in
    fy = (torch.arange(0, r0, r0 / h) % 1)[:h].to(device)
    fx = (torch.arange(0, r1, r1 / w) % 1)[:w].to(device)
    cy = (torch.arange(h, device=device) // cell[0]).clamp_(max=r0 - 1)
    cx = (torch.arange(w, device=device) // cell[1]).clamp_(max=r1 - 1)
    py, px = fy[:, None].expand(h, w), fx[None, :].expand(h, w)

    def corner(dy, dx):  # gradient of the cell's corner (dy, dx) dotted with the offset from that corner
        g = grad[(cy + dy)[:, None], (cx + dx)[None, :]]
        return torch.stack(((py - dy) * g[..., 0], (px - dx) * g[..., 1]), dim=-1).sum(dim=-1)

    ty, tx = _fade(py), _fade(px)
    top = torch.lerp(corner(0, 0), corner(1, 0), ty)
    bottom = torch.lerp(corner(0, 1), corner(1, 1), ty)
    return math.sqrt(2) * torch.lerp(top, bottom, tx)


def perlin_2d_octaves(shape, res, octaves=1, persistence=0.5, device="cpu"):
    noise = torch.zeros(tuple(int(v) for v in shape), device=device)
    frequency, amplitude = 1, 1
    for _ in range(octaves):
        noise += amplitude * perlin_2d(shape, (frequency * res[0], frequency * res[1]), device=device)
        frequency *= 2
        amplitude *= persistence
    return noise


class NoiseTextureLerpSampler(base.Sampler):
    """sample() -> [3, H, W]: colour_a .. colour_b blended by normalised multi-octave Perlin noise (:63-102)."""

    def __init__(self, color_a, color_b, texture_shape, eval_step_size: float = 0.01, device=torch.device("cuda")) -> None:
        super().__init__(torch.tensor([0.0], device=device), torch.tensor([1.0], device=device), eval_step_size, device)
        self._color_a, self._color_b = color_a, color_b
        self._texture_shape = texture_shape

    def sample_train(self):
        lattice = 2 ** random.randint(1, 6)
        octaves = random.randint(1, 4)
        persistence = random.uniform(0.1, 2.0)
        t = perlin_2d_octaves(self._texture_shape, (lattice, lattice), octaves, persistence, device=self._device)
        t = (t - t.min()) / (t.max() - t.min())
        a = self._color_a.to(self._device).reshape(3, 1, 1).expand(3, *t.shape)
        b = self._color_b.to(self._device).reshape(3, 1, 1).expand(3, *t.shape)
        return torch.lerp(a, b, t.unsqueeze(0).expand(3, *t.shape))

    def sample_eval(self):  # (the reference draws a random texture in eval mode too, :100-102)
        return self.sample_train()

    def sample(self):
        return self.sample_train()
