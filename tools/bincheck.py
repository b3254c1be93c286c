"""Self-check of the tile bins (needs the -DFFX_BINCHECK build: tools/build_variant_lib.sh bincheck -DFFX_BINCHECK; FFX_LIB=...): every walk a
pixel's packet makes through the bins is repeated on the tree inside the same kernel and compared lane by lane — closest hits by primitive
and distance, shadow walks by their occlusion flags — over random poses.  Prints the number of waves with a differing lane per walk kind and
the first case of each.

    FFX_LIB=fireflies_amd/csrc/_stats/libffx_hip_bincheck.so python tools/bincheck.py [vocalfold|colon] [poses] [spp]
"""
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _lib, mi, workloads  # noqa: E402


def _f(bits):
    import struct

    return struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    poses = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    lib = _lib.api().lib
    if not hasattr(lib, "ffx_debug_bincheck"):
        raise SystemExit("needs the -DFFX_BINCHECK build (FFX_LIB=...)")
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    buf = (C.c_ulonglong * 24)()
    lib.ffx_debug_bincheck(buf, 1)
    tot = [0, 0, 0]
    for p in range(poses):
        torch.manual_seed(500 + p)
        random.seed(500 + p)
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=spp, seed=p).torch()
        torch.cuda.synchronize()
        lib.ffx_debug_bincheck(buf, 1)
        b = list(buf)
        for j, base in enumerate((0, 8, 16)):
            tot[j] += b[base]
        if b[0] or b[8] or b[16]:
            def px(v):
                return (v & 0xFFFF, (v >> 16) & 0xFFFF, (v >> 32) & 0xFF, (v >> 40) & 1)
            print(f"pose {p}: primary {b[0]} (first: pixel/lane {px(b[2])[:3]}, prim bins/tree {b[3] & 0xFFFFFFFF}/{b[3] >> 32}, slot {b[4] & 0xFFFFFFFF}/{b[4] >> 32}, t {_f(b[5] & 0xFFFFFFFF)!r}/{_f(b[5] >> 32)!r}); "
                  f"projector shadow {b[8]} (first {px(b[10])}, tree's occluder slot {b[11]}); spot shadow {b[16]} (first {px(b[18])}, tree's occluder slot {b[19]})")
    print(f"{which}: {poses} poses x {spp} spp: waves with a lane where bins and tree disagree — primary {tot[0]}, projector shadow {tot[1]}, spot shadow {tot[2]}")


if __name__ == "__main__":
    main()
