"""The gaussian reconstruction filter's cost on the render: renders of the same poses with mi.Scene.rfilter = "box" and
"gaussian", one after the other on one stream (nothing overlaps: the figures are kernel times), per pose.  Under rocprofv3 --kernel-trace --stats
the per-kernel split (k_render_fwd_pk<..., true> = the render with the filter's fold, k_rf_gather) comes out of the same run.

    python tools/rftime.py [vocalfold|colon] [spp]
"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if which == "vocalfold" else 256)
    if which == "vocalfold":
        wl = workloads.vocalfold(device="cuda", width=512, height=512, grid=16)
    else:
        wl = workloads.colon(device="cuda", width=1024, height=1024, grid=32)
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    fp16 = which == "colon"
    res = {}
    for mode in ("box", "gaussian", "box", "gaussian"):
        wl.mi_scene.rfilter = mode
        ms = []
        for seed in range(6):
            torch.manual_seed(seed)
            random.seed(seed)
            wl.ff_scene.randomize()
            mi.render(wl.mi_scene, spp=spp, seed=seed, fp16=fp16).torch()  # (the pre-pass of this pose runs here)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(8):
                mi.render(wl.mi_scene, spp=spp, seed=seed + k, fp16=fp16).torch()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b) / 8)
        res.setdefault(mode, []).append(sum(ms) / len(ms))
        print(f"{mode:9s} {sum(ms) / len(ms):.4f} ms per render  per pose {[round(m, 4) for m in ms]}", flush=True)
    print({k: [round(x, 4) for x in v] for k, v in res.items()})


if __name__ == "__main__":
    main()
