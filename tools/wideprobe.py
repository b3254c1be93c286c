"""A/B probe of the packet walks (binary: FFX_WIDE=0, 64-wide: default) on K7 / K8: time per launch and, with
a -DFFX_STATS library (tools/build_stats_lib.sh; FFX_LIB=...), steps per walk."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _lib, mi, workloads  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def stats():
    lib = _lib.api().lib
    if not hasattr(lib, "ffx_debug_stats"):
        return None
    buf = (C.c_ulonglong * 32)()
    lib.ffx_debug_stats(buf, 1)
    out = {}
    for base, kind in ((0, "closest"), (4, "any")):
        w = max(buf[base], 1)
        out[kind] = dict(walks=buf[base], steps=round(buf[base + 1] / w, 2), tris=round(buf[base + 2] / w, 2), stage2=round(buf[base + 3] / w, 2),
                         leafy=round(buf[base + 8] / w, 2), inner_hit=round(buf[base + 9] / w, 2), pops=round(buf[base + 10] / w, 2),
                         max_steps_tris=(buf[11 if base == 0 else 15] >> 32, buf[11 if base == 0 else 15] & 0xFFFFFFFF))
        e = 16 if base == 0 else 20
        f = 24 if base == 0 else 28
        out[kind].update(walks_gt16tris=buf[e], tris_beyond16=buf[e + 1], walks_gt64tris=buf[e + 2], tris_beyond64=buf[e + 3],
                         walks_gt12steps=buf[f], steps_beyond12=buf[f + 1], generic_walks=buf[f + 2], total_tris=buf[base + 2], total_steps=buf[base + 1])
    return out


def timers():
    lib = _lib.api().lib
    buf = (C.c_ulonglong * 32)()
    lib.ffx_debug_timers(buf, 1)
    return list(buf)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    geom = wl.mi_scene.geom
    cam = wl.mi_scene.camera_struct(0)
    has_stats = hasattr(_lib.api().lib, "ffx_debug_stats")
    for wide in ("0", "1"):
        os.environ["FFX_WIDE"] = wide
        for spp in (1, 64):
            for jit in (0, 1):
                if has_stats:
                    stats()
                    geom.trace_primary(cam, spp, jit, 3, want_ids=False)
                    torch.cuda.synchronize()
                    print(f"wide={wide} K7 spp={spp} jitter={jit}:", stats()["closest"])
                else:
                    ms = timeit(lambda: geom.trace_primary(cam, spp, jit, 3, want_ids=False))
                    print(f"wide={wide} K7 spp={spp} jitter={jit}: {ms:.3f} ms")
        if has_stats:
            stats()
            mi.render(wl.mi_scene, spp=64, seed=1)
            torch.cuda.synchronize()
            st = stats()
            print(f"wide={wide} K8:", st)
        if hasattr(_lib.api().lib, "ffx_debug_timers"):
            timers()
            mi.render(wl.mi_scene, spp=64, seed=1)
            torch.cuda.synchronize()
            t = timers()
            npx = 262144.0
            names = ["fetch+test", "exact tests", "select+push", "pop", "setup"]
            for base, kind in ((0, "closest"), (8, "any")):
                print(f"wide={wide} K8 cycles per pixel, {kind} walks:", {n: round(t[base + i] / npx) for i, n in enumerate(names)})
            print(f"wide={wide} K8 cycles per pixel: raygen {t[16] / npx:.0f}, shade_sample_pk (3 walks + shading) {t[17] / npx:.0f}")
        else:
            for sh in (True, False):
                wl.mi_scene.shadows = sh
                wl.mi_scene._sd_cache = None
                ms = timeit(lambda: mi.render(wl.mi_scene, spp=64, seed=1))
                print(f"wide={wide} K8 64 spp shadows={sh}: {ms:.3f} ms")
            wl.mi_scene.shadows = True
            wl.mi_scene._sd_cache = None


if __name__ == "__main__":
    main()
