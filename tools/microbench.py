#!/usr/bin/env python3
"""Per-kernel timings (HIP events on the launch stream) and achieved GB/s against the algorithmic
bytes of DESIGN.md §5.  Run on an MI355X:  python tools/microbench.py > profiles/<tag>_microbench.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import ops, workloads  # noqa: E402

PEAK = 8000.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def row(name, ms, nbytes, note=""):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": name, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4), "note": note}


def main():
    dev = "cuda"
    out = []
    rng = np.random.default_rng(0)
    for N, S in ((64, 500), (256, 500), (1024, 1024)):
        pts = torch.from_numpy((rng.random((N, 2)) * 0.9 + 0.05).astype(np.float32)).to(dev)
        T = S * S
        if N * T * 4 <= 2**31:
            ms = timeit(lambda: ops.splat_dense_fwd(pts, 10.0, S, S), iters=10)
            out.append(row(f"splat_dense_fwd N={N} {S}x{S}", ms, 8 * N + 4 * N * T, "rasterize_points: HBM-write bound"))
            g = torch.rand((N, S, S), device=dev)
            ms = timeit(lambda: ops.splat_dense_bwd(pts, 10.0, S, S, g), iters=10)
            r = int(np.sqrt(10.3 * 10.0)) + 2
            out.append(row(f"splat_dense_bwd N={N} {S}x{S}", ms, 16 * N + 4 * N * (2 * r + 1) ** 2, "reads only the non-zero footprint of gout"))
            del g
        for mode in ("sum", "softor"):
            ms = timeit(lambda: ops.splat_fwd(pts, 10.0, mode, -1, S, S))
            out.append(row(f"splat_fwd {mode} N={N} {S}x{S}", ms, 8 * N + 4 * T))
            tex = ops.splat_fwd(pts, 10.0, mode, -1, S, S)
            gt = torch.rand_like(tex)
            ms = timeit(lambda: ops.splat_bwd(pts, 10.0, mode, -1, S, S, tex, gt))
            out.append(row(f"splat_bwd {mode} N={N} {S}x{S}", ms, 16 * N + 4 * N * 23 * 23))
        img = torch.rand((S, S), device=dev)
        ms = timeit(lambda: ops.blur_fwd(img))
        out.append(row(f"blur_fwd 5x5 {S}x{S}", ms, 8 * T))
        ms = timeit(lambda: ops.blur_bwd(img))
        out.append(row(f"blur_bwd 5x5 {S}x{S}", ms, 8 * T))
    rays = torch.rand((1024, 3), device=dev) - 0.5
    KF = np.eye(4, dtype=np.float32)
    KF[3, 3], KF[3, 2] = 0.0, 1.0
    ms = timeit(lambda: ops.project_rays_fwd(rays, KF))
    out.append(row("project_rays_fwd N=1024", ms, 24 * 1024, "launch-latency bound"))

    wl = workloads.vocalfold(device=dev, entity_device="cpu")
    geom = wl.mi_scene.geom
    V = sum(m.frames.shape[1] for m in wl.data.meshes)
    F, NN = geom.n_tris, geom.info.n_nodes
    G = 12 * V + 12 * F + 32 * NN
    xf = torch.eye(4).repeat(2, 1, 1)
    ms = timeit(lambda: geom.update(xf))
    out.append(row("scene_update (K5+K6, one launch)", ms, 12 * V + 48 * F + 128 * NN, f"V={V} F={F} nodes={NN}"))
    cam = wl.mi_scene.camera_struct(0)
    W = H = 512
    for spp, jit in ((1, 0), (64, 1)):
        ms = timeit(lambda: geom.trace_primary(cam, spp, jit, 3), iters=10)
        out.append(row(f"trace_primary 512x512 spp={spp}", ms, G + 12 * W * H * spp, "t + shape + prim per sample; rays/s = %.3g" % (W * H * spp / (ms * 1e-3))))
        ms = timeit(lambda: geom.trace_primary(cam, spp, jit, 3, want_ids=False), iters=10)
        out.append(row(f"trace_primary 512x512 spp={spp} (t only)", ms, G + 4 * W * H * spp, "rays/s = %.3g" % (W * H * spp / (ms * 1e-3))))
    with torch.no_grad():
        tex = workloads.build_texture(wl).contiguous()
    sd = wl.mi_scene.scene_desc(tex_channels=1)
    alb = wl.mi_scene.albedo
    T = 250000
    ms = timeit(lambda: geom.render_fwd(sd, alb, tex.unsqueeze(-1), 64, 1), iters=10)
    out.append(row("render_fwd 512x512x64spp shadows", ms, G + 4 * T + 12 * W * H, "samples/s = %.3g" % (W * H * 64 / (ms * 1e-3))))
    g = torch.rand((H, W, 3), device=dev)
    ms = timeit(lambda: geom.render_bwd(sd, alb, 64, 1, g), iters=10)
    out.append(row("render_bwd 512x512x64spp shadows", ms, G + 8 * T + 12 * W * H, "includes the gtex memset"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
