"""Debug helper: per-walk node-step / triangle-test counts of the packet traversal.
Needs a library built with -DFFX_STATS (FFX_LIB=... python tools/walkstats.py)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _lib, workloads  # noqa: E402


def main():
    from fireflies_amd import mi

    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    diffuse = len(sys.argv) > 3 and sys.argv[3] == "diffuse"  # (the Lambert kernel; default: the scene's principled material)
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu", principled=not diffuse)
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    lib = _lib.api().lib
    buf = (C.c_ulonglong * 32)()
    for seed in range(3):
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=spp, seed=seed)
        torch.cuda.synchronize()
        lib.ffx_debug_stats(buf, 1)
    names = ["walks", "node_steps", "tri_tests", "tri_stage2"]
    extra = ["leafy_steps", "both_hit_steps", "pops"]
    for base, kind in ((0, "closest"), (4, "any")):
        w = max(buf[base], 1)
        d = {n: round(buf[base + i] / w, 2) for i, n in enumerate(names[1:], 1)}
        d.update({n: round(buf[base + 8 + i] / w, 2) for i, n in enumerate(extra)})
        print(kind, "walks", buf[base], "per walk:", d)


if __name__ == "__main__":
    main()
