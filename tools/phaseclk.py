#!/usr/bin/env python3
"""Where K8's wave time goes, from shader-clock stamps inside the kernel (a -DFFX_TIMERS build of the library:
tools/build_variant_lib.sh timers -DFFX_TIMERS; FFX_LIB=fireflies_amd/csrc/_stats/libffx_hip_timers.so python tools/phaseclk.py).
Each wave adds the s_memtime ticks (the shader clock counter) it spends in a phase to a per-phase total; with 7-8 waves per SIMD taking
turns on the issue ports, a phase's share of the wave time is (to first order) its share of the issue slots.  Prints ticks per
pixel and the share of the whole wave, nested as the code nests.  The stamps perturb the kernel (s_memtime + s_waitcnt per stamp:
the build runs ~10 % slower) — shares, not absolute times.

    python tools/phaseclk.py [vocalfold|colon] [spp] [poses]
"""
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _lib, mi, workloads  # noqa: E402

SLOTS = {25: "prologue (tile / pixel coordinates, constants)", 16: "ray generation (jitter hash, camera ray)", 22: "primary walk (closest hit)",
         18: "light terms at the hit (normal, projector / spot geometry, texel probe)", 19: "projector shadow walk", 20: "spot shadow walk",
         21: "BSDF of the lit samples + footprint indices", 23: "epilogue (cache fold, texture gather, 64-lane sums, store)"}
WALK = ["fetch + 64 box tests", "exact triangle tests", "nearest child + push", "pop", "packet set-up"]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    poses = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lib = _lib.api().lib
    if not hasattr(lib, "ffx_debug_timers"):
        raise SystemExit("needs a -DFFX_TIMERS library (see the docstring)")
    res = 512 if which == "vocalfold" else 1024
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", width=res, height=res)
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    buf = (C.c_ulonglong * 32)()
    torch.manual_seed(1000)
    random.seed(1000)
    tot = [0] * 32
    for i in range(poses):
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=spp, seed=i)  # (the first render also pays set-up work: timers are reset before each)
        torch.cuda.synchronize()
        lib.ffx_debug_timers(buf, 1)
        mi.render(wl.mi_scene, spp=spp, seed=i)
        torch.cuda.synchronize()
        lib.ffx_debug_timers(buf, 1)
        for k in range(32):
            tot[k] += buf[k]
    npx = float(res * res * poses)
    whole = tot[24] / npx
    print(f"{which} {res}x{res} x {spp} spp, {poses} poses: {whole:.1f} s_memtime ticks of wave time per pixel")
    acc = 0.0
    for k, name in SLOTS.items():
        v = tot[k] / npx
        acc += v
        print(f"  {name:78s} {v:8.1f}  {100 * v / whole:5.1f} %")
        if k in (22, 20):  # the sub-phases of the TREE walks: only printed when a packet walked the tree (with the tile bins hardly any does)
            base = 0 if k == 22 else 8
            sub = tot[base:base + 5]
            lab = "closest-hit walks" if k == 22 else "any-hit walks (both emitters)"
            for j in (4, 0, 1, 2, 3):
                if 100 * sub[j] / npx / whole >= 0.05:  # (the handful of packets that fall back to the tree would print a row of zeros)
                    print(f"      {lab} (tree fallback): {WALK[j]:40s} {sub[j] / npx:8.1f}  {100 * sub[j] / npx / whole:5.1f} %")
    print(f"  {'(sum of the phases)':78s} {acc:8.1f}  {100 * acc / whole:5.1f} %")


if __name__ == "__main__":
    main()
