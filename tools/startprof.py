#!/usr/bin/env python3
"""Start-up transient of bench.py's render bracket: host time per step and GPU completion interval per step
for the first steps after a synchronize (what a 20-step bracket sees).  Run on an MI355X."""
import gc
import os
import random
import sys
import time

import torch

os.environ.setdefault("FFX_RENDER_STREAMS", "1")  # (per-step GPU intervals are read from events on the CURRENT stream: renders must run there)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402

dev = torch.device("cuda", 0)
wl = workloads.vocalfold(device=dev, width=512, height=512, grid=16, entity_device=sys.argv[2] if len(sys.argv) > 2 else "cuda")
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
torch.manual_seed(1000)
random.seed(1000)


split = []


def step(i):
    t0 = time.perf_counter()
    wl.ff_scene.randomize()
    t1 = time.perf_counter()
    r = mi.render(wl.mi_scene, spp=64, seed=1000 + i)
    split.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
    return r


warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = os.environ.get("FFX_START_MODE", "")
if "gcfirst" in mode:
    gc.collect()
    gc.disable()
for i in range(warm):
    step(i)
if "gcfirst" not in mode:
    gc.collect()
    gc.disable()
torch.cuda.synchronize()
if "spin" in mode:  # keep the host core busy after the blocking synchronize
    t_end = time.perf_counter() + 0.005
    while time.perf_counter() < t_end:
        pass
if "dummy" in mode:  # one trivial launch first: does the FIRST submission after idle carry the cost?
    t0 = time.perf_counter()
    dummy = torch.empty(16, device=dev).fill_(1.0)
    print(f"dummy launch host {1e3 * (time.perf_counter() - t0):.3f} ms")
n = 30
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
t00 = time.perf_counter()
evs[0].record()
pr = None
if os.environ.get("FFX_PROF") == "first":
    import cProfile
    pr = cProfile.Profile()
for i in range(n):
    t0 = time.perf_counter()
    if pr is not None and i == 0:
        pr.enable()
    step(warm + i)
    if pr is not None and i == 0:
        pr.disable()
    evs[i + 1].record()
    host.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t00)
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print(f"warm {warm}: wall {wall:.2f} ms for {n} steps = {wall / n:.4f} ms/step; first 20: {sum(gpu[:20]) / 20:.4f} gpu-interval avg")
print("host ms :", " ".join(f"{t:.2f}" for t in host))
print("gpu  ms :", " ".join(f"{t:.2f}" for t in gpu))
print("randomize/render host ms:", " ".join(f"{a:.2f}/{b:.2f}" for a, b in split[warm:warm + 8]))
if pr is not None:
    import pstats
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
elif os.environ.get("FFX_PROF"):
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    step(1000)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
