"""Point-pattern optimisation (the reference ships this file EMPTY; BASELINE configs[1]):
64-point laser pattern, 512x512, 64 spp, one randomised scene sample per step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    wl = workloads.vocalfold(grid=8, entity_device="cpu")
    opt = PatternOptimizer(wl.mi_scene, wl.ff_scene, wl.laser, sigma=wl.sigma, tex_size=wl.tex_size, spp=64, lr=1e-3, samples_per_step=1)
    for i in range(steps):
        out = opt.step()
        if i % 10 == 0:
            print(f"step {i:4d}  loss {float(out['loss']):.6f}")
    wl.laser.save("optimised_pattern.yaml")
    print("saved optimised_pattern.yaml")
