#!/usr/bin/env python3
"""Do back-to-back launches of K8 leave a tail that a second stream could fill?  2N renders of one pose on ONE stream against N + N on two
streams (same blob, read-only; separate images)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402

wl = workloads.vocalfold(device="cuda", width=512, height=512, grid=16)
ms, g = wl.mi_scene, wl.mi_scene.geom
with torch.no_grad():
    tex = workloads.build_texture(wl).contiguous().unsqueeze(-1).contiguous()
wl.ff_scene.randomize()
sd = ms.scene_desc(tex_channels=1)
mats = ms.materials_arg(sd)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 40


def run(streams):
    for _ in range(4):
        g.render_fwd(sd, mats, tex, 64, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for i in range(2 * N):
        with torch.cuda.stream(streams[i % len(streams)]):
            g.render_fwd(sd, mats, tex, 64, 3)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (2 * N)


for rep in range(5):
    a2 = run([s1, s2]); a1 = run([s1]); b2 = run([s1, s2]); b1 = run([s1])
    print(f"two streams {a2:7.1f}  one stream {a1:7.1f}  two {b2:7.1f}  one {b1:7.1f} us per render")
