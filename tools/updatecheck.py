"""The native params.update() (ffx_scene_step_h, ABI 8) against the key-by-key path over workload variants the test suite does not carry: two scenes of
one configuration, one held on the Python path — scene description byte for byte, material rows, transforms, parameter values, image — for the
vocal fold with a diffuse material and without shadow rays, the colon with its mucosa randomisation and with a diffuse material.
    python tools/updatecheck.py"""
import os, sys, random, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, warnings
warnings.simplefilter("ignore")
from fireflies_amd import workloads, mi
def run(name, make):
    ws = []
    for native in (False, True):
        wl = make()
        wl.ff_scene.native_update = native
        with torch.no_grad():
            wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
        ws.append(wl)
    a, b = ws
    for k in range(12):
        for wl in ws:
            torch.manual_seed(40 + k); random.seed(40 + k)
            wl.ff_scene.randomize()
        sa, sb = a.mi_scene.scene_desc(tex_channels=1), b.mi_scene.scene_desc(tex_channels=1)
        assert C.string_at(C.addressof(sa), C.sizeof(sa)) == C.string_at(C.addressof(sb), C.sizeof(sb)), (name, k)
        assert np.array_equal(a.mi_scene._albedo_host, b.mi_scene._albedo_host) and torch.equal(a.mi_scene._xforms, b.mi_scene._xforms)
        ia, ib = mi.render(a.mi_scene, spp=4, seed=k).torch(), mi.render(b.mi_scene, spp=4, seed=k).torch()
        assert torch.equal(ia, ib) and float(ia.sum()) > 0, (name, k)
        for key in a.params.keys():
            va, vb = a.params[key], b.params[key]
            if isinstance(va, float): assert float(va) == float(vb), (name, k, key)
    print(name, "ok; python scene", a.mi_scene.update_paths, "native scene", b.mi_scene.update_paths, b.mi_scene.update_fallbacks)
run("vocalfold diffuse", lambda: workloads.vocalfold(width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24), principled=False))
run("colon principled", lambda: workloads.colon(width=96, height=96, tex=128, grid=6, n_around=32, n_along=64))
run("colon diffuse", lambda: workloads.colon(width=96, height=96, tex=128, grid=6, n_around=32, n_along=64, principled=False))
run("vocalfold no shadows", lambda: workloads.vocalfold(width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24), shadows=False))
