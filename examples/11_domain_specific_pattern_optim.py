"""Domain-specific pattern optimisation (EMPTY in the reference; BASELINE configs[3]): 32 randomised
scene samples per step, sharded over the GPUs of one node, ONE all-reduce of the pattern gradient per
step.  Launch:  torchrun --standalone --nproc-per-node 8 examples/11_domain_specific_pattern_optim.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fireflies_amd import dist, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rank, world, local = dist.init()
    torch.cuda.set_device(local)
    wl = workloads.vocalfold(device=torch.device("cuda", local), grid=16, entity_device="cpu")
    opt = PatternOptimizer(wl.mi_scene, wl.ff_scene, wl.laser, sigma=wl.sigma, tex_size=wl.tex_size, spp=64, lr=1e-3, samples_per_step=32)
    for i in range(steps):
        out = opt.step()
        if rank == 0 and i % 5 == 0:
            print(f"step {i:4d}  loss {float(out['loss']):.6f}  ({world} GPUs x {32 // world} samples)")
    if rank == 0:
        wl.laser.save("optimised_pattern_domain.yaml")
