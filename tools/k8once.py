"""renders the vocal-fold workload a few times (for rocprofv3 passes): python tools/k8once.py [n] [shadows 0/1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = workloads.vocalfold(device="cuda", entity_device="cpu")
if len(sys.argv) > 2 and sys.argv[2] == "0":
    wl.mi_scene.shadows = False
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
for i in range(n):
    mi.render(wl.mi_scene, spp=64, seed=i)
torch.cuda.synchronize()
