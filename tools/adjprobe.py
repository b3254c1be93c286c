#!/usr/bin/env python3
"""K8 variants on one pose of the gradient bracket's scene (64-point pattern), back to back: plain forward, cache-writing forward (+ K9),
fused forward + adjoint with and without the <gimg, img> partial sums."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import _abi, ops, workloads  # noqa: E402

wl = workloads.vocalfold(device="cuda", width=512, height=512, grid=8)
ms, g = wl.mi_scene, wl.mi_scene.geom
with torch.no_grad():
    tex = workloads.build_texture(wl).contiguous().unsqueeze(-1).contiguous()
wl.ff_scene.randomize()
sd = ms.scene_desc(tex_channels=1)
mats = ms.materials_arg(sd)
gimg = torch.zeros((512, 512, 3), device="cuda")
gimg[..., 1] = -1.0 / 512**2
gtex = torch.zeros_like(tex)
dot = torch.zeros(_abi.ADJOINT_DOT_SLOTS, device="cuda")
cache = torch.zeros(ops.render_cache_bytes_sd(sd, 64), dtype=torch.uint8, device="cuda")


def t(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


for sparse in (True, False):
    print(f"sparse adjoint = {sparse}")
    print(f"  plain forward                         {t(lambda: g.render_fwd(sd, mats, tex, 64, 3)):7.1f} us")
    print(f"  cache-writing forward                 {t(lambda: g.render_fwd(sd, mats, tex, 64, 3, cache=cache, sparse_adjoint=sparse)):7.1f} us")
    print(f"  cache-writing forward + K9            {t(lambda: (g.render_fwd(sd, mats, tex, 64, 3, cache=cache, sparse_adjoint=sparse), g.render_bwd_cached(sd, mats, cache, 64, gimg, out=gtex))):7.1f} us")
    print(f"  fused forward + adjoint, no dot       {t(lambda: g.render_fwd_adjoint(sd, mats, tex, 64, 3, gimg, out=gtex, sparse_adjoint=sparse)):7.1f} us")
    print(f"  fused forward + adjoint, dot slots    {t(lambda: g.render_fwd_adjoint(sd, mats, tex, 64, 3, gimg, out=gtex, dot_out=dot, sparse_adjoint=sparse)):7.1f} us")
