"""Tile bins against the tree walks on K8: kernel time with FFX_BINS=1 / 0 on the same poses, the bins' own
status (entries, capacity, list lengths per tile) and — with a -DFFX_STATS library (tools/build_stats_lib.sh; FFX_LIB=...) — what a
pixel's packet does: chunk steps, exact tests and fall-backs to the tree per walk.

    python tools/binstats.py [vocalfold|colon] [spp]
"""
import ctypes as C
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _lib, mi, workloads  # noqa: E402

NT = 16384
OFF_STARTS, OFF_ENTRIES = 64, ((64 + 4 * (NT + 16) + 4 * NT + 63) // 64) * 64


def bins_status(geom):
    """header + list-length statistics of the three grids of the blob the next render reads"""
    info, blob = geom.info, geom.blob
    torch.cuda.synchronize()
    out = []
    for a, name in enumerate(("camera", "projector", "spot")):
        base = int(info.off_bins) + a * int(info.bins_stride)
        hdr = blob[base: base + 64].cpu().numpy().view(np.uint32)
        starts = blob[base + OFF_STARTS: base + OFF_STARTS + 4 * (NT + 1)].cpu().numpy().view(np.uint32)
        total = int(hdr[1])
        d = np.diff(starts.astype(np.int64))  # (list starts are non-decreasing over the grid's tiles; the words behind them are zero)
        neg = np.nonzero(d < 0)[0]
        ln = d[: int(neg[0])] if len(neg) else d
        if ln.size == 0:
            ln = np.zeros(1, np.int64)
        out.append({"apex": name, "ok": int(hdr[0]), "entries": total, "capacity": int(hdr[2]), "tiles_with_entries": int((ln > 0).sum()),
                    "mean_list": float(ln[ln > 0].mean()) if (ln > 0).any() else 0.0, "p95_list": float(np.percentile(ln[ln > 0], 95)) if (ln > 0).any() else 0.0,
                    "max_list": int(ln.max())})
    return out


def stats48():
    lib = _lib.api().lib
    if not hasattr(lib, "ffx_debug_stats48"):
        return None
    buf = (C.c_ulonglong * 48)()
    lib.ffx_debug_stats48(buf, 1)
    return list(buf)


def k8_ms(wl, spp, poses, fp16=False):
    ms = []
    for seed in poses:
        torch.manual_seed(seed)
        random.seed(seed)
        wl.ff_scene.randomize()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(4):
            mi.render(wl.mi_scene, spp=spp, seed=seed + k, fp16=fp16).torch()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) / 4)
    return ms


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if which == "vocalfold" else 256)
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    os.environ["FFX_RENDER_STREAMS"] = "1"
    poses = list(range(100, 108)) if which == "vocalfold" else [100, 101]
    img = {}
    for bins in ("1", "0", "1", "0"):
        os.environ["FFX_BINS"] = bins
        wl.mi_scene._sd_cache = None
        ms = k8_ms(wl, spp, poses, fp16=which == "colon")
        print(f"FFX_BINS={bins}: K8 {np.mean(ms):.4f} ms per render (4 back-to-back renders of each of {len(poses)} poses, incl. the pre-pass of the first); per pose", [round(m, 4) for m in ms])
        img[bins] = mi.render(wl.mi_scene, spp=spp, seed=5).torch().clone()
    same = torch.equal(img["1"], img["0"])
    print("images with and without bins identical:", same, "" if same else f"max diff {float((img['1'].float() - img['0'].float()).abs().max()):.3e}")
    os.environ["FFX_BINS"] = "1"
    wl.mi_scene._sd_cache = None
    mi.render(wl.mi_scene, spp=spp, seed=1).torch()
    for row in bins_status(wl.mi_scene.geom):
        print(row)
    st = stats48()
    if st is not None:
        mi.render(wl.mi_scene, spp=spp, seed=1).torch()
        torch.cuda.synchronize()
        st = stats48()
        for base, kind in ((32, "closest (camera)"), (40, "any (emitters)")):
            w = max(st[base], 1)
            print(f"bins, {kind}: walks {st[base]}, chunk steps per walk {st[base + 1] / w:.2f}, exact tests {st[base + 2] / w:.2f}, second barycentric {st[base + 3] / w:.2f}")
        print(f"any-hit candidates within reach (t-range pre-check passed): {st[46] / max(st[40], 1):.2f} per walk")
        print(f"fall-backs to the tree: primary {st[36]}, projector shadow {st[44]}, spot shadow {st[45]}")
        print(f"tree walks that ran: closest {st[0]} ({st[1] / max(st[0], 1):.2f} steps), any {st[4]} ({st[5] / max(st[4], 1):.2f} steps)")


if __name__ == "__main__":
    main()
