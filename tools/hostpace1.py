"""Host pace of a ONE-sample gradient step (the bench's gradient brackets): small film and mesh, so that the GPU is never the limit — what the host needs per
step for the linear loss (fused render + pattern launch) and for an L1 loss (cache-writing render, loss launch, K9, pattern launch).   python tools/hostpace1.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer, image_l1_loss  # noqa: E402

KW = dict(width=64, height=56, tex=96, frames=5, n_fold=20, tube=(20, 24))
for kind in ("linear", "l1"):
    for merged in ("1", "0"):
        os.environ["FFX_PATTERN_STEP"] = merged
        wg = workloads.vocalfold(grid=8, **KW)
        kw = {}
        if kind == "l1":
            with torch.no_grad():
                kw["loss_fn"] = image_l1_loss(mi.render(wg.mi_scene, spp=4, seed=99).torch().clone())
        opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=4, samples_per_step=1, base_seed=7, **kw)
        for _ in range(20):
            opt.step()
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            opt.step()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{kind:6s} FFX_PATTERN_STEP={merged}: host issues a step in {1e6 * t_issue / n:.1f} us; with the GPU's tail {1e6 * t_all / n:.1f} us", flush=True)
