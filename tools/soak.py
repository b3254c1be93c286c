#!/usr/bin/env python3
"""Soak: thousands of randomise + render steps and gradient steps on the default (principled) workload — finite images,
finite losses, no growth of the device allocation, steady step time.  Run on an MI355X:  python tools/soak.py [renders] [grad steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

n_r = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_g = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
wl = workloads.vocalfold()
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
acc = torch.zeros((), device="cuda", dtype=torch.float64)
bad = torch.zeros((), device="cuda", dtype=torch.int64)
mem0 = None
t0 = time.perf_counter()
for i in range(n_r):
    wl.ff_scene.randomize()
    img = mi.render(wl.mi_scene, spp=64, seed=i).torch()
    if i % 16 == 0:
        acc += img.double().mean()
        bad += (~torch.isfinite(img)).sum()
    if i == 200:
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
    if i % 1000 == 999:
        torch.cuda.synchronize()
        print(f"renders {i + 1}: {1e3 * (time.perf_counter() - t0) / (i + 1):.4f} ms/step, mean radiance {float(acc) / ((i // 16) + 1):.5f}, non-finite {int(bad)}", flush=True)
torch.cuda.synchronize()
print("allocated bytes after 200 / after all renders:", mem0, torch.cuda.memory_allocated())
wg = workloads.vocalfold(grid=8)
opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
t0 = time.perf_counter()
losses = []
for i in range(n_g):
    out = opt.step()
    if i % 100 == 99:
        losses.append(float(out["loss"]))
    if i == 200:
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
    if i % 500 == 499:
        torch.cuda.synchronize()
        print(f"grad steps {i + 1}: {1e3 * (time.perf_counter() - t0) / (i + 1):.4f} ms/step, loss {losses[-1]:.5f}", flush=True)
torch.cuda.synchronize()
rays = wg.laser._rays.detach()
print("allocated bytes after 200 / after all grad steps:", mem0, torch.cuda.memory_allocated())
print("losses finite:", all(l == l and abs(l) < 1e9 for l in losses), "rays finite:", bool(torch.isfinite(rays).all()), "unit rays:",
      float((rays.norm(dim=1) - 1).abs().max()))
