"""How many triangles the pre-pass proves "clear" (nothing can shadow them from an emitter: ffx_bins.hip k_bin_clear) and how many pixels of a
render skip an emitter's any-hit stage because of it — default workload, a few random poses.  The pixel shares need the -DFFX_STATS build
(tools/build_stats_lib.sh; counters 38 / 39 = packets that skipped the projector's / the spot's stage, 40 / 44 = shadow packets walked).

    python tools/clearstats.py [vocalfold|colon]
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    wl = (workloads.vocalfold(device="cuda", width=512, height=512, grid=16) if which == "vocalfold" else workloads.colon(device="cuda", width=1024, height=1024, grid=32))
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    gd = wl.mi_scene.geom
    for pose in range(4):
        torch.manual_seed(pose)
        random.seed(pose)
        wl.ff_scene.randomize()
        mi.render(wl.mi_scene, spp=64 if which == "vocalfold" else 16, seed=pose).torch()
        torch.cuda.synchronize()
        info, blob = gd.info, gd.blob
        F = int(info.n_tris)
        w = blob[int(info.off_gn): int(info.off_gn) + 16 * F].cpu().numpy().view(np.uint32).reshape(F, 4)[:, 3]
        ok = (w & 0x0FFFFFFF) != 0
        shape = (w & 0x0FFFFFFF) - 1
        out = []
        for a, name in ((1, "projector"), (2, "spot")):
            bit = (w >> (27 + a)) & 1
            per_shape = {int(s): round(float(bit[ok & (shape == s)].mean()), 3) for s in np.unique(shape[ok])}
            out.append(f"{name} {float(bit[ok].mean()):.3f} per shape {per_shape}")
        print(f"pose {pose}: clear triangles: " + "; ".join(out), flush=True)


if __name__ == "__main__":
    main()
