"""Host pace of a multi-sample gradient step (BASELINE configs[3]: 32 scene samples per step): the optimiser's loop with 1 spp, so that the GPU is
never the limit — per scene sample the time the host needs to randomise, re-fit, launch the render + adjoint.    python tools/hostprof_step.py [S] [small]
"small": the same entities, samplers and keys over a 64x56 film and ~2 k triangles — at full size the device's chain per pose (re-fit -> count -> scan -> fill ->
render, ~200 us of dependent launches whatever the spp) sets the pace of a 1-spp loop and the host's launch calls wait for it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402
from fireflies_amd.optim import PatternOptimizer  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
KW = dict(width=64, height=56, tex=96, frames=5, n_fold=20, tube=(20, 24)) if (len(sys.argv) > 2 and sys.argv[2] == "small") else {}
for spp in (1, 64):
    wg = workloads.vocalfold(grid=8, **KW)
    opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=spp, samples_per_step=S, base_seed=7)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        opt.step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{S} samples per step, {spp} spp: host issues a step in {1e3 * t_issue / n:.3f} ms = {1e6 * t_issue / n / S:.1f} us per scene sample; "
          f"with the GPU drained {1e3 * t_all / n:.3f} ms per step = {1e6 * t_all / n / S:.1f} us per sample; pushes {wg.mi_scene.update_paths}", flush=True)

if os.environ.get("FFX_HP_PROFILE") == "1":
    import cProfile
    import pstats

    wg = workloads.vocalfold(grid=8, **KW)
    opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=1, samples_per_step=S, base_seed=7)
    for _ in range(5):
        opt.step()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        opt.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumtime").print_stats(32)
    st.sort_stats("tottime").print_stats(22)
