#!/usr/bin/env python3
"""Is a short bracket slower because of WHEN it runs or because of WHICH poses it renders?  Renders the bench's pose sequence
(seed 1000, 90 steps) several times in one process, GPU time per step from events, and prints the mean per window of ten steps.
  pass 0   first use                      pass 1   straight after pass 0 (same poses: any difference is warm-up, not the poses)
  pass 2   after 0.3 s of idle GPU        pass 3   after a gc.collect() (what bench.py does in front of a bracket)
  pass 4   after 0.3 s of idle, then 48 back-to-back renders of one pose (26 ms of the same kernel) in front of the loop
  pass 5   after 0.3 s of idle, then 4 renders with the per-lane kernels (40 ms of a latency-bound kernel) in front of the loop
Result on MI355X (profiles/r3_posecost.txt): the first ~40 renders after an idle phase run ~6 % slower — whatever ran before the idle
phase; a short burst of the same kernel in front of the loop does / does not remove it (see the file)."""
import gc
import os
import random
import sys
import time

import torch

os.environ.setdefault("FFX_RENDER_STREAMS", "1")  # (per-step GPU intervals are read from events on the CURRENT stream: renders must run there)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402

wl = workloads.vocalfold(device="cuda", width=512, height=512, grid=16)
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
for p in range(6):
    if p in (2, 4, 5):
        torch.cuda.synchronize()
        time.sleep(0.3)
    if p == 3:
        gc.collect()
    if p == 4:
        for k in range(48):
            mi.render(wl.mi_scene, spp=64, seed=k)
    if p == 5:
        os.environ["FFX_TRAVERSAL"] = "lane"
        for k in range(4):
            mi.render(wl.mi_scene, spp=64, seed=k)
        os.environ.pop("FFX_TRAVERSAL")
    torch.manual_seed(1000)
    random.seed(1000)
    n = 90
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=64, seed=1000 + i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    g = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(f"pass {p}: " + " ".join(f"{sum(g[a:a + 10]) / 10:.3f}" for a in range(0, n, 10)) + f"   (steps 5..24: {sum(g[5:25]) / 20:.4f} ms)")
