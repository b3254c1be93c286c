"""cProfile of the host side of a render step (randomize + mi.render), entity device from argv[1]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402

dev = sys.argv[1] if len(sys.argv) > 1 else "cpu"
wl = workloads.vocalfold(device="cuda", entity_device=dev)
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()


def step(i):
    wl.ff_scene.randomize()
    mi.render(wl.mi_scene, spp=64, seed=i)


for i in range(20):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
