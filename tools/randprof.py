"""host cost of ff_scene.randomize() by phase, entity device cpu vs cuda (GPU otherwise idle / busy rendering)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import entity, mi, workloads  # noqa: E402

for dev in ("cpu", "cuda"):
    wl = workloads.vocalfold(device="cuda", entity_device=dev)
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    sc = wl.ff_scene
    for busy in (False, True):
        for _ in range(5):
            sc.randomize()
        torch.cuda.synchronize()
        t = {"draw": 0.0, "fetch": 0.0, "apply": 0.0, "render_launch": 0.0}
        n = 50
        for i in range(n):
            t0 = time.perf_counter()
            batch = entity.DrawBatch()
            side = sc._draw_stream()
            if side is None:
                drawn = sc._draw_all(batch)
            else:
                with torch.cuda.stream(side):
                    drawn = sc._draw_all(batch)
            t1 = time.perf_counter()
            values = sc._fetch(batch)
            t2 = time.perf_counter()
            sc._apply(drawn, values)
            t3 = time.perf_counter()
            if busy:
                mi.render(wl.mi_scene, spp=64, seed=i)
            t4 = time.perf_counter()
            t["draw"] += t1 - t0
            t["fetch"] += t2 - t1
            t["apply"] += t3 - t2
            t["render_launch"] += t4 - t3
        torch.cuda.synchronize()
        print(dev, "busy" if busy else "idle", {k: round(v / n * 1e6, 1) for k, v in t.items()}, "us per step")
