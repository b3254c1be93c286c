"""K8 below 64 samples per pixel: k_render_fwd_blk (several pixels per wave, the default below 33 spp) against the pixel-per-wave kernel
(FFX_RENDER_BLOCKS=0) on the same poses, each timed alone with HIP events (device drained, nothing overlapping).  512x512 vocal fold.
    python tools/lowspp.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import workloads  # noqa: E402


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(n):
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return sum(ms) / len(ms)


wl = workloads.vocalfold(device="cuda")
ms, geom = wl.mi_scene, wl.mi_scene.geom
with torch.no_grad():
    tex = workloads.build_texture(wl).contiguous()
tex3 = tex[..., 1:2].contiguous() if tex.dim() == 3 else tex.unsqueeze(-1).contiguous()
rows = {}
for pose in range(4):
    torch.manual_seed(pose)
    random.seed(pose)
    wl.ff_scene.randomize()
    sd = ms.scene_desc(tex_channels=1)
    mats = ms.materials_arg(sd)
    for spp in (1, 2, 4, 8, 16, 24, 32, 48, 64):
        os.environ.pop("FFX_RENDER_BLOCKS", None)
        t_blk = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7))
        os.environ["FFX_RENDER_BLOCKS"] = "0"
        t_one = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7))
        os.environ.pop("FFX_RENDER_BLOCKS", None)
        r = rows.setdefault(spp, [0.0, 0.0])
        r[0] += t_blk / 4
        r[1] += t_one / 4
print("box film:\nspp   pixel blocks   pixel per wave   (ms, mean of 4 poses)")
for spp, (a, b) in rows.items():
    print(f"{spp:3d}   {a:10.4f}   {b:12.4f}   x{b / a:5.2f}")
# the gaussian film (the filtered render: kernel + gather), the film of every scene the reference loads
ms.rfilter = "gaussian"
rows = {}
for pose in range(4):
    torch.manual_seed(pose)
    random.seed(pose)
    wl.ff_scene.randomize()
    sd = ms.scene_desc(tex_channels=1)
    mats = ms.materials_arg(sd)
    for spp in (1, 4, 8, 10, 12, 16, 32, 64):
        os.environ.pop("FFX_RENDER_BLOCKS", None)
        t_blk = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7))
        os.environ["FFX_RENDER_BLOCKS"] = "0"
        t_one = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7))
        os.environ.pop("FFX_RENDER_BLOCKS", None)
        r = rows.setdefault(spp, [0.0, 0.0])
        r[0] += t_blk / 4
        r[1] += t_one / 4
print("gaussian film (kernel + gather):\nspp   pixel blocks   pixel per wave")
for spp, (a, b) in rows.items():
    print(f"{spp:3d}   {a:10.4f}   {b:12.4f}   x{b / a:5.2f}")
