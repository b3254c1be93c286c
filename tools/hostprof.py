import sys, time, random; sys.path.insert(0,'.')
import torch
from fireflies_amd import workloads, mi
for edev in ("cuda","cpu"):
    wl = workloads.vocalfold(device="cuda", entity_device=edev)
    with torch.no_grad():
        tex = workloads.build_texture(wl).contiguous()
    wl.params["tex.data"] = tex
    for i in range(3):
        wl.ff_scene.randomize(); mi.render(wl.mi_scene, spp=64, seed=i)
    torch.cuda.synchronize()
    tr=tu=tm=0
    N=20
    for i in range(N):
        torch.manual_seed(i); random.seed(i)
        t0=time.perf_counter()
        wl.ff_scene.randomize_list(wl.ff_scene._meshes); wl.ff_scene.randomize_list(wl.ff_scene._lights); wl.ff_scene.randomize_list(wl.ff_scene._materials)
        t1=time.perf_counter()
        wl.ff_scene.update_meshes(); wl.ff_scene.update_lights(); wl.ff_scene.update_materials(); wl.ff_scene._mitsuba_params.update()
        t2=time.perf_counter()
        mi.render(wl.mi_scene, spp=64, seed=i)
        t3=time.perf_counter()
        torch.cuda.synchronize()
        tr+=t1-t0; tu+=t2-t1; tm+=t3-t2
    print(edev, "randomize %.3f ms, update %.3f ms, render-enqueue %.3f ms" % (tr/N*1e3, tu/N*1e3, tm/N*1e3))
