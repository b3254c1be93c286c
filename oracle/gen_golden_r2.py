#!/usr/bin/env python3
"""Round-2 golden vectors from the REFERENCE's own code (same rules as gen_golden.py: runs only in the build
container, the output under tests/golden/ is data, no reference source text is stored):

  g10_lines_grad.npz   rasterize_lines forward and its autograd gradient w.r.t. the segments
                       (fireflies/graphics/rasterization.py:107-153; the optimisation loop at :645-743 builds the
                       segments from a leaf, so the in-place scaling at :122-123 is legal there — done the same way here)
  g11_bridson.npz      sampling.poisson.bridson under np.random.seed (fireflies/sampling/poisson.py:16-116, which
                       draws from the GLOBAL numpy generator) and Laser.generate_blue_noise_rays
                       (fireflies/projection/laser.py:95-145) on top of it
"""
import os

import numpy as np
import torch

from gen_golden import OUT, import_reference, perspective_projection


def main():
    R, M, S, E, P = import_reference()
    import fireflies.sampling.poisson as RP

    cpu = torch.device("cpu")
    g10 = {}
    torch.manual_seed(31)
    for tag, n, size, sigma in (("a", 4, (20, 20), 3.0), ("b", 6, (32, 32), 10.0), ("c", 3, (48, 48), 40.0)):
        leaf = (torch.rand(n, 2, 2) * 0.8 + 0.1).requires_grad_(True)
        lines = leaf * 1.0  # non-leaf: the reference scales its argument in place
        tsz = torch.tensor(size)
        out = R.rasterize_lines(lines, sigma, tsz, device=cpu)
        w = torch.cos(torch.arange(out.numel(), dtype=torch.float32) * 0.37).reshape(out.shape)
        (out * w).sum().backward()
        g10[f"{tag}_lines"] = leaf.detach().numpy()
        g10[f"{tag}_size"] = np.asarray(size)
        g10[f"{tag}_sigma"] = np.float32(sigma)
        g10[f"{tag}_out"] = out.detach().numpy()
        g10[f"{tag}_w"] = w.numpy()
        g10[f"{tag}_glines"] = leaf.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g10_lines_grad.npz"), **g10)

    g11 = {}
    for tag, seed, shape, r in (("const", 5, (40, 30), 6.25), ("small", 9, (24, 24), 3.0)):
        np.random.seed(seed)
        n, pts = RP.bridson(np.ones(shape) * r)
        g11[f"{tag}_seed"], g11[f"{tag}_shape"], g11[f"{tag}_radius"] = np.int64(seed), np.asarray(shape), np.float64(r)
        g11[f"{tag}_n"], g11[f"{tag}_pts"] = np.int64(n), np.asarray(pts, np.float64)
    # spatially varying radius, k = 12, and the normal-distributed variant
    yy, xx = np.meshgrid(np.arange(36), np.arange(28), indexing="ij")
    rmap = 2.5 + 0.1 * xx + 0.05 * yy
    np.random.seed(13)
    n, pts = RP.bridson(rmap, k=12)
    g11["vary_map"], g11["vary_n"], g11["vary_pts"] = rmap, np.int64(n), np.asarray(pts, np.float64)
    np.random.seed(14)
    n, pts = RP.bridson(np.ones((30, 30)) * 4.0, k=20, radiusType="normDist")
    g11["norm_n"], g11["norm_pts"] = np.int64(n), np.asarray(pts, np.float64)
    K = torch.from_numpy(perspective_projection(32, 24, 40.0, 0.01, 100.0))
    np.random.seed(21)
    rays = P.Laser.generate_blue_noise_rays(32, 24, 16, K, device=cpu)
    g11["bn_K"], g11["bn_rays"] = K.numpy(), rays.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g11_bridson.npz"), **g11)
    for f in ("g10_lines_grad.npz", "g11_bridson.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
