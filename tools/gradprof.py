#!/usr/bin/env python3
"""Host-side cost of one pattern-gradient step, section by section (enqueue time, no device syncs
inside the step), next to the synchronous wall time per step.  Run on an MI355X."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wg = workloads.vocalfold(device=dev, grid=8, entity_device=sys.argv[1] if len(sys.argv) > 1 else "cuda")  # (bench.py's default: cuda)
    opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=64, samples_per_step=1, base_seed=7)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        opt.step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"step: host enqueue {1e3 * t_enq / n:.3f} ms, wall {1e3 * t_all / n:.3f} ms")
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        opt.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
