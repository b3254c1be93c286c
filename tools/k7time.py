"""K7 (ffx_trace_primary) time per call for a range of samples per pixel: python tools/k7time.py  (FFX_K7_PPW_LOG2 caps the pixels per wave)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
geom = wl.mi_scene.geom
cam = wl.mi_scene.camera_struct(0)


def t(spp, jit):
    for _ in range(3):
        geom.trace_primary(cam, spp, jit, 3, want_ids=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        geom.trace_primary(cam, spp, jit, 3, want_ids=False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20


print(which, "cap", os.environ.get("FFX_K7_PPW_LOG2"), " ".join(f"{s}spp {t(s, 1 if s > 1 else 0):.4f}" for s in (1, 2, 4, 8, 16, 32, 64)))
