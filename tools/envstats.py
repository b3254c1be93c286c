"""Round 6: the emitters' ENVELOPES (k_bin_env + the proof in bins_shadow) against the plain any-hit stage on K8: kernel time with
FFX_ENVELOPE=3 / 0 on the same poses, image identity, the pre-pass alone with and without the envelope launch, and — with a -DFFX_STATS
library (tools/build_stats_lib.sh; FFX_LIB=...) — the share of shadow packets the proof settles.

    python tools/envstats.py [vocalfold|colon] [spp]
"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads  # noqa: E402
from tools.binstats import k8_ms, stats48  # noqa: E402


def prepass_ms(wl, n=40):
    """randomize + params.update() (re-fit, apex records, bins [, envelopes]) on an idle GPU: device time per pose"""
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for k in range(n):
        wl.ff_scene.randomize()
        wl.mi_scene.flush() if hasattr(wl.mi_scene, "flush") else None
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, (time.perf_counter() - t0) * 1e3 / n


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vocalfold"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if which == "vocalfold" else 256)
    wl = (workloads.vocalfold if which == "vocalfold" else workloads.colon)(device="cuda", entity_device="cpu")
    with torch.no_grad():
        wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
    os.environ["FFX_RENDER_STREAMS"] = "1"
    poses = list(range(100, 108)) if which == "vocalfold" else [100, 101]
    img = {}
    for env in ("3", "0", "3", "0", "2", "1"):
        os.environ["FFX_ENVELOPE"] = env
        wl.mi_scene._sd_cache = None
        ms = k8_ms(wl, spp, poses, fp16=which == "colon")
        print(f"FFX_ENVELOPE={env}: K8 {np.mean(ms):.4f} ms per render (4 back-to-back renders of each of {len(poses)} poses, incl. the pre-pass of the first); per pose", [round(m, 4) for m in ms])
        img[env] = mi.render(wl.mi_scene, spp=spp, seed=5).torch().clone()
    for env in ("0", "2", "1"):
        same = torch.equal(img["3"], img[env])
        print(f"images with FFX_ENVELOPE=3 and ={env} identical:", same, "" if same else f"max diff {float((img['3'].float() - img[env].float()).abs().max()):.3e}")
    for env in ("3", "0", "3", "0"):
        os.environ["FFX_ENVELOPE"] = env
        wl.mi_scene._sd_cache = None
        torch.manual_seed(1)
        random.seed(1)
        dev, host = prepass_ms(wl)
        print(f"FFX_ENVELOPE={env}: randomize + update alone {dev:.4f} ms of device time per pose (host {host:.4f} ms)")
    os.environ["FFX_ENVELOPE"] = "3"
    wl.mi_scene._sd_cache = None
    st = stats48()
    if st is not None:
        mi.render(wl.mi_scene, spp=spp, seed=1).torch()
        torch.cuda.synchronize()
        stats48()
        mi.render(wl.mi_scene, spp=spp, seed=1).torch()
        torch.cuda.synchronize()
        st = stats48()
        # 40: any-hit walks served by the bins' exact stage; 37 / 47: projector / spot packets the envelope settled; 44 / 45: fall-backs to the tree
        print(f"shadow packets: envelope settled projector {st[37]}, spot {st[47]}; exact any-hit stages {st[40]} (chunk steps {st[41] / max(st[40], 1):.2f}, "
              f"candidates {st[42] / max(st[40], 1):.2f}); tree fall-backs projector {st[44]}, spot {st[45]}")
        tot = st[37] + st[47] + st[40] + st[44] + st[45]
        print(f"share of shadow packets settled by the envelope: {(st[37] + st[47]) / max(tot, 1):.3f}")


if __name__ == "__main__":
    main()
