#!/usr/bin/env python3
"""Diagnostic: GPU time per gradient step (one event per step) after a long render bracket, as in
bench.py --steps 150.  Run on an MI355X."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 150
wl = workloads.vocalfold(device=dev, grid=16, entity_device="cpu")
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
for i in range(pre):
    torch.manual_seed(i)
    random.seed(i)
    wl.ff_scene.randomize()
    mi.render(wl.mi_scene, spp=64, seed=i)
torch.cuda.synchronize()
wg = workloads.vocalfold(device=dev, grid=8, entity_device="cpu")
opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    opt.step()
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print("mean %.3f ms" % (sum(ms) / n))
print(" ".join(f"{t:.2f}" for t in ms))
