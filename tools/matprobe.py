"""K8 time of the vocal-fold render for: a diffuse scene (3-float rows), material rows with every row Lambert (model 0:
the cost of the material-row kernel's plumbing and occupancy), and principled rows (the BSDF itself).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fireflies_amd import mi, workloads


def k8(wl, n=60):
    with torch.no_grad():
        tex = workloads.build_texture(wl).contiguous()
    wl.params["tex.data"] = tex
    for i in range(10):
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=64, seed=i)
    torch.cuda.synchronize()
    ev = []
    for i in range(n):
        wl.ff_scene.randomize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        mi.render(wl.mi_scene, spp=64, seed=100 + i)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms[: n // 2]) / (n // 2)


print("diffuse            %.4f ms" % k8(workloads.vocalfold(principled=False)))
wl = workloads.vocalfold()
print("principled         %.4f ms" % k8(wl))
wl = workloads.vocalfold(randomize=False)
wl.mi_scene._albedo_host[:, 3] = 0.0
wl.mi_scene._albedo_stale = True  # (the device table / the scene description pick the edited host rows up)
wl.mi_scene._sd_cache = None
print("rows, all Lambert  %.4f ms" % k8(wl))
wl = workloads.vocalfold(randomize=False)
print("principled, fixed  %.4f ms" % k8(wl))
wl = workloads.vocalfold(principled=False, randomize=False)
print("diffuse, fixed     %.4f ms" % k8(wl))
