"""The pre-pass of a packet render in isolation (ffx_apex_prepare: apex records + tile bins, two launches): time per call on an otherwise
idle GPU, with and without the bins (FFX_BINS=0: the apex records alone).

    python tools/binprobe.py [vocalfold|colon]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import ops, workloads  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    wl.ff_scene.randomize()
    geom = wl.mi_scene.geom
    torch.cuda.synchronize()
    for bins in ("1", "0", "1"):
        os.environ["FFX_BINS"] = bins
        wl.mi_scene._sd_cache = None
        sd = wl.mi_scene.scene_desc(tex_channels=1)
        blob = geom.blob
        args = (ops._dev(blob, torch.uint8, "blob"), C.byref(geom.info), C.byref(sd))
        for _ in range(5):
            geom._call("ffx_apex_prepare", *args, ops._stream(geom._didx))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        a.record()
        for _ in range(n):
            geom._call("ffx_apex_prepare", *args, ops._stream(geom._didx))
        b.record()
        torch.cuda.synchronize()
        print(f"FFX_BINS={bins}: pre-pass {1e3 * a.elapsed_time(b) / n:.1f} us per call (back to back, idle GPU)")


if __name__ == "__main__":
    main()
