"""Phase timers (perf_counter, no profiler overhead) of the host side of Scene.randomize() + mi.render on the
vocal-fold workload: where the 0.37 ms per step go."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import entity, mi, workloads  # noqa: E402

wl = workloads.vocalfold(device="cuda", entity_device="cuda")
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
sc = wl.ff_scene
acc = {}
SPP = int(os.environ.get("FFX_HP_SPP", "64"))  # (a small value keeps the GPU ahead of the host: pure host time per step)


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(i):
    t = time.perf_counter()
    batch = entity.DrawBatch()
    drawn = sc._draw_all(batch)
    t = tick("draw_all", t)
    values = batch.fetch()
    t = tick("fetch/_values", t)
    ents, tickets, picks = drawn
    for e, tk in zip(ents, tickets):
        e._compose(tk, values)
    t = tick("compose", t)
    for m, pick in zip(sc._meshes, picks):
        m._pending_pick = pick
    sc.update_meshes()
    for m in sc._meshes:
        del m._pending_pick
    t = tick("update_meshes", t)
    sc.update_camera()
    sc.update_projector()
    sc.update_lights()
    sc.update_materials()
    t = tick("update_cam/proj/lights/mats", t)
    sc._mitsuba_params.update()
    t = tick("params.update (refit launch)", t)
    mi.render(wl.mi_scene, spp=SPP, seed=i).torch()
    t = tick("mi.render (launch)", t)


for i in range(30):
    step(i)
torch.cuda.synchronize()
acc.clear()
N = 300
t00 = time.perf_counter()
SYNC = os.environ.get("FFX_HP_SYNC") == "1"  # an idle GPU in front of every step: what the FIRST step of a bracket pays
for i in range(N):
    if SYNC:
        torch.cuda.synchronize()
    step(i)
tot = time.perf_counter() - t00
torch.cuda.synchronize()
for k, v in acc.items():
    print(f"{k:36s} {1e6 * v / N:7.1f} us")
print(f"{'total host per step':36s} {1e6 * tot / N:7.1f} us")
# the plain loop (no phase timers), and — FFX_HP_PROFILE=1 — where its time goes by function
import random

t00 = time.perf_counter()
for i in range(N):
    sc.randomize()
    mi.render(wl.mi_scene, spp=SPP, seed=i).torch()
tot = time.perf_counter() - t00
torch.cuda.synchronize()
print(f"{'ff_scene.randomize() + mi.render':36s} {1e6 * tot / N:7.1f} us per step (plain loop)")
if os.environ.get("FFX_HP_PROFILE") == "1":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for i in range(N):
        sc.randomize()
        mi.render(wl.mi_scene, spp=SPP, seed=i).torch()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumtime").print_stats(40)
