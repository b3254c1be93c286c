"""cProfile of the host side of PatternOptimizer.step (what bench.py's gradient bracket runs), 300 steps."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

wg = workloads.vocalfold(device="cuda", grid=8)
opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
for _ in range(20):
    opt.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    opt.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(40)
