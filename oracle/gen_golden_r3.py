#!/usr/bin/env python3
"""Round-3 golden vectors from the REFERENCE's own code (same rules as gen_golden.py: runs only in the build container,
the output under tests/golden/ is data, no reference source text is stored):

  g13_noise_texture.npz     sampling.NoiseTextureLerpSampler.sample() under torch / random seeds
                            (fireflies/sampling/noise_texture_lerp.py:8-102) — the texture the reference's dataset loop
                            assigns to `mat-Mucosa.brdf_0.base_color.data` every iteration (main.py:138-153)
  g14_camera_out_of_bounds.npz  Laser.randomize_camera_out_of_bounds on a CPU seed (fireflies/projection/laser.py:233-249)
"""
import os
import random

import numpy as np
import torch

from gen_golden import OUT, import_reference, perspective_projection


def main():
    R, M, S, E, P = import_reference()
    from fireflies.sampling.noise_texture_lerp import NoiseTextureLerpSampler

    cpu = torch.device("cpu")
    g13 = {}
    # (the lattice goes up to 64 * 2^3 cells per axis, so the texture must be at least 512 wide — the reference uses 1024^2;
    # the fixture keeps every 8th texel plus the mean and the sum of squares of the whole texture)
    for tag, seed, shape in (("a", 3, (512, 512)), ("b", 11, (1024, 512)), ("c", 12, (512, 512))):
        torch.manual_seed(seed)
        random.seed(seed)
        ca, cb = torch.rand(3), torch.rand(3)
        smp = NoiseTextureLerpSampler(color_a=ca, color_b=cb, texture_shape=shape, device=cpu)
        g13[f"{tag}_seed"], g13[f"{tag}_shape"] = np.int64(seed), np.asarray(shape)
        g13[f"{tag}_color_a"], g13[f"{tag}_color_b"] = ca.numpy(), cb.numpy()
        for k in range(2):  # the generators keep running between the two draws
            t = smp.sample().numpy()
            g13[f"{tag}_tex{k}_sub"] = t[:, ::8, ::8].copy()
            g13[f"{tag}_tex{k}_mean"], g13[f"{tag}_tex{k}_sq"] = np.float64(t.astype(np.float64).mean()), np.float64((t.astype(np.float64) ** 2).sum())
    np.savez_compressed(os.path.join(OUT, "g13_noise_texture.npz"), **g13)

    K = torch.from_numpy(perspective_projection(500, 500, 30.0, 0.01, 100.0))
    tr = E.Transformable("projector", cpu)
    g14 = {"K": K.numpy()}
    rays = P.Laser.generate_uniform_rays(0.0275 * 18 / 8, 8, 8, device=cpu)
    laser = P.Laser(tr, rays.clone(), K, 30.0, 0.01, 100.0, device=cpu)
    torch.manual_seed(17)
    ndc = torch.rand(64, 3) * 3.0 - 1.5  # camera-space NDC of the laser points: about a third leaves [-1, 1]^2
    g14["rays_before"], g14["ndc"] = laser._rays.clone().numpy(), ndc.numpy()
    torch.manual_seed(23)
    laser.randomize_camera_out_of_bounds(ndc)
    g14["rays_after"] = laser._rays.clone().numpy()
    # nothing out of bounds: untouched
    inside = torch.rand(64, 3) * 1.0 - 0.5
    laser2 = P.Laser(tr, rays.clone(), K, 30.0, 0.01, 100.0, device=cpu)
    laser2.randomize_camera_out_of_bounds(inside)
    g14["inside"], g14["rays_inside_after"] = inside.numpy(), laser2._rays.clone().numpy()
    np.savez_compressed(os.path.join(OUT, "g14_camera_out_of_bounds.npz"), **g14)
    for f in ("g13_noise_texture.npz", "g14_camera_out_of_bounds.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
