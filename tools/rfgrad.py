"""The pieces of a gradient step through the gaussian film, each timed alone on the same poses (HIP events, device
drained, nothing overlapping): the filtered forward, the filtered forward that also stores its adjoint's per-sample records (dense and with
FFX_RENDER_SPARSE_ADJOINT), the adjoint from those records (k_render_bwd_cached_filtered), the fused forward + adjoint launch of a linear
loss, and the re-tracing adjoint.  64-point pattern (the gradient bracket's), 512x512, 64 spp.

    python tools/rfgrad.py [grid] [spp]
"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import ops, workloads  # noqa: E402


def timed(fn, n=6):
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


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    wl = workloads.vocalfold(device="cuda", width=512, height=512, grid=grid)
    ms, geom = wl.mi_scene, wl.mi_scene.geom
    with torch.no_grad():
        tex = workloads.build_texture(wl).contiguous()
    tex3 = tex[..., 1:2].contiguous() if tex.dim() == 3 else tex.unsqueeze(-1).contiguous()
    ms.rfilter = "gaussian"
    H = W = 512
    gimg = torch.randn((H, W, 3), device="cuda").sign_() / (3.0 * H * W)
    rows = {}
    for pose in range(4):
        torch.manual_seed(pose)
        random.seed(pose)
        wl.ff_scene.randomize()
        sd = ms.scene_desc(tex_channels=1)
        mats = ms.materials_arg(sd)
        cache = torch.empty(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
        gtex = torch.zeros((sd.proj.tex_h, sd.proj.tex_w, 1), device="cuda")
        r = rows.setdefault(pose, {})
        r["fwd_filtered"] = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7))
        r["fwd_cache_dense"] = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7, cache=cache))
        lit_dense = int((cache[64:64 + 8 * W * H].view(W * H, 8)[:, 6:8].view(torch.int16) != 0).sum())
        r["k9f_dense"] = timed(lambda: geom.render_bwd_cached(sd, mats, cache, spp, gimg, out=gtex, seed=7))
        r["fwd_cache_sparse"] = timed(lambda: geom.render_fwd(sd, mats, tex3, spp, 7, cache=cache, sparse_adjoint=True))
        lit_sparse = int((cache[64:64 + 8 * W * H].view(W * H, 8)[:, 6:8].view(torch.int16) != 0).sum())
        r["k9f_sparse"] = timed(lambda: geom.render_bwd_cached(sd, mats, cache, spp, gimg, out=gtex, seed=7))
        # ... and back to back behind a render (the clocks a step sees; the drained figures above include the ramp of an idle GPU)
        geom.render_fwd(sd, mats, tex3, spp, 7, cache=cache, sparse_adjoint=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            geom.render_bwd_cached(sd, mats, cache, spp, gimg, out=gtex, seed=7)
        b.record()
        torch.cuda.synchronize()
        r["k9f_sparse_b2b"] = a.elapsed_time(b) / 30
        if os.environ.get("RFGRAD_ONLY_K9"):
            print(f"pose {pose}: lit pixels dense {lit_dense} sparse {lit_sparse}  " + "  ".join(f"{k} {v:.4f}" for k, v in r.items()), flush=True)
            continue
        r["fused_linear_sparse"] = timed(lambda: geom.render_fwd_adjoint(sd, mats, tex3, spp, 7, gimg, out=gtex, sparse_adjoint=True))
        r["retrace"] = timed(lambda: geom.render_bwd(sd, mats, spp, 7, gimg))
        print(f"pose {pose}: lit pixels dense {lit_dense} sparse {lit_sparse}  " + "  ".join(f"{k} {v:.4f}" for k, v in r.items()), flush=True)
        del cache
    keys = list(rows[0])
    print("mean ms: " + "  ".join(f"{k} {sum(rows[p][k] for p in rows) / len(rows):.4f}" for k in keys))


if __name__ == "__main__":
    main()
