import sys, os; sys.path.insert(0,'.')
import torch, numpy as np
from fireflies_amd import workloads
from tools.microbench import timeit
wl = workloads.vocalfold(device="cuda", entity_device="cpu")
geom = wl.mi_scene.geom
print("max_depth", geom.info.max_depth, "nodes", geom.info.n_nodes, "levels", geom.info.n_levels)
cam = wl.mi_scene.camera_struct(0)
for spp in (1,2,4,16,64):
    for jit in (0,1):
        ms = timeit(lambda: geom.trace_primary(cam, spp, jit, 3, want_ids=False), iters=5, warm=1)
        print(f"spp={spp} jitter={jit}: {ms:.3f} ms  {512*512*spp/ms/1e6:.1f} Mrays/ms-scale")
# random incoherent rays
n=262144
o=torch.zeros((n,3),device="cuda"); o[:,2]=1.5
d=torch.randn((n,3),device="cuda"); d[:,2]=d[:,2].abs()+0.5; d=d/d.norm(dim=1,keepdim=True)
ms=timeit(lambda: geom.trace_rays(o,d), iters=5, warm=1); print("trace_rays random 262144:", ms)
