"""Where the host time of one loop step goes on the native randomiser path (round 4): wall-clock wrappers (perf_counter_ns, ~0.3 us each) around
the functions of `ff_scene.randomize()` + `mi.render(...)`, nested names indented.  1 spp by default so that the GPU is never the limit.

    python tools/hostprof4.py [spp]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, scene_desc, workloads  # noqa: E402

SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# "small": the same entities, samplers and parameter keys over a 64x56 film and ~2 k triangles — the host's work per step is the same, the device's
# chain (re-fit -> count -> scan -> fill -> render: ~200 us of dependent launches at full size, whatever the spp) no longer sets the pace
SMALL = len(sys.argv) > 2 and sys.argv[2] == "small"
KW = dict(width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24)) if SMALL else {}
wl = workloads.vocalfold(device="cuda", entity_device="cuda", **KW)
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
sc, ms = wl.ff_scene, wl.mi_scene
acc, cnt = {}, {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter_ns()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0) + time.perf_counter_ns() - t0
            cnt[label] = cnt.get(label, 0) + 1

    setattr(obj, name, w)


wrap(sc, "_randomize_native", "  randomize: native draw + chains (C)")
wrap(sc, "_apply_native", "  randomize: _apply_native (all of the below)")
wrap(sc, "_push_native", "    _push_native (ABI 8: one call for the sample; the map is told afterwards)")
wrap(ms, "step_native", "      mi.Scene.step_native (ffx_scene_step_h + blob bookkeeping)")
wrap(ms.geom, "update_native", "        geom.update_native")
wrap(sc, "update_camera", "    update_camera")
wrap(sc, "update_projector", "    update_projector")
wrap(sc, "update_lights", "    update_lights")
wrap(sc, "update_materials", "    update_materials")
wrap(ms._params, "update", "    params.update()")
wrap(ms, "_apply", "      mi.Scene._apply")
wrap(ms, "scene_desc", "        scene_desc() [also called by render]")
wrap(scene_desc, "scene_desc", "          scene_desc.scene_desc (build)")
wrap(ms.geom, "update", "        geom.update")
wrap(ms.geom, "_update_into", "          _update_into (ffx_scene_update_h)")
wrap(ms.geom, "_prepare_apex", "          _prepare_apex (ffx_apex_prepare: records + bins)")
wrap(ms.geom, "render_fwd", "  render: geom.render_fwd")
wrap(ms, "_render_stream", "  render: _render_stream")


def loop(n):
    t0 = time.perf_counter()
    for i in range(n):
        sc.randomize()
        mi.render(ms, spp=SPP, seed=i).torch()
    return time.perf_counter() - t0


loop(50)
torch.cuda.synchronize()
acc.clear(); cnt.clear()
N = 400
tot = loop(N)
torch.cuda.synchronize()
print(f"{1e6 * tot / N:7.1f} us per step (wrapped loop, {SPP} spp); geometry pushes {ms.update_paths}")
for k, v in acc.items():
    print(f"{k:70s} {1e-3 * v / N:7.1f} us  ({cnt[k] / N:.0f} calls)")
# ... and the loop as a script runs it (no wrappers)
wl2 = workloads.vocalfold(device="cuda", entity_device="cuda", **KW)
with torch.no_grad():
    wl2.params["tex.data"] = workloads.build_texture(wl2).contiguous()
sc, ms = wl2.ff_scene, wl2.mi_scene
loop(50)
torch.cuda.synchronize()
t = min(loop(N) for _ in range(3))
torch.cuda.synchronize()
print(f"{1e6 * t / N:7.1f} us per step, unwrapped (best of 3 x {N}); geometry pushes {ms.update_paths}")
