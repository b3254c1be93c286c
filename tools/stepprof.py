#!/usr/bin/env python3
"""Per-step host time of PatternOptimizer.step without device syncs (what bench.py's gradient bracket
does), and the overall wall time.  Run on an MI355X."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

dev = torch.device("cuda", 0)
wg = workloads.vocalfold(device=dev, grid=8, entity_device=os.environ.get("FFX_ENTITY_DEVICE", "cuda"))
opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
if len(sys.argv) > 2 and sys.argv[2] == "timing":  # bench.py's per-launch event pairs
    wg.mi_scene.geom.timing = []
ts = []
t00 = time.perf_counter()
SYNC = os.environ.get("FFX_HP_SYNC") == "1"  # idle GPU in front of every step: the unthrottled host time of a step
for i in range(n):
    if SYNC:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.step()
    ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("wall per step %.3f ms" % (1e3 * (time.perf_counter() - t00) / n))
print(" ".join(f"{t:.2f}" for t in ts))
