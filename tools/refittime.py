#!/usr/bin/env python3
"""K5+K6 alone: time of ffx_scene_update_h per call with nothing else on the GPU (synchronous single-blob mode), over the treelet size of
the refit plan (FFX_TREELET_TRIS, read by the host builder) and, for comparison, the level-by-level launches (FFX_REFIT=levels).

    python tools/refittime.py [vocalfold|colon] [treelet sizes ...]
"""
import os
import sys

os.environ["FFX_ASYNC_UPDATE"] = "0"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402


def time_update(wl, iters=60):
    ms, g = wl.mi_scene, wl.mi_scene.geom
    for _ in range(5):
        g.update(ms._xforms, ms._offs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.update(ms._xforms, ms._offs)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    sizes = [int(v) for v in sys.argv[2:]] or [256, 512, 1024, 2048, 4096, 8192]
    make = workloads.vocalfold if which == "vocalfold" else workloads.colon
    res = 512 if which == "vocalfold" else 1024
    for t in sizes:
        os.environ["FFX_TREELET_TRIS"] = str(t)
        wl = make(device="cuda", width=res, height=res)
        info = wl.mi_scene.geom.info
        us = time_update(wl)
        os.environ["FFX_REFIT"] = "levels"
        us_l = time_update(wl)
        os.environ.pop("FFX_REFIT")
        print(f"{which}: treelets of <= {t:5d} triangles: {info.n_treelets:5d} treelets, one launch {us:7.1f} us per update; level launches {us_l:7.1f} us", flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
