#!/usr/bin/env python3
"""Fuzz: packet K7 / K8 against the CPU oracle over many random poses (run on an MI355X).
Counts rays whose primitive differs and — the failure a non-conservative box test would cause — rays the
oracle hits but the GPU misses; and the worst per-pixel radiance difference."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import ops, scene_desc, scenes  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n_pose = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    which = sys.argv[2] if len(sys.argv) > 2 else "vocalfold"
    rng = np.random.default_rng(123)
    if which == "colon":  # the 524,288-triangle scene of config 5 at a small film
        sc = scenes.colon(width=96, height=80, tex=64)
    else:
        sc = scenes.vocalfold(width=96, height=80, tex=64, frames=6, n_fold=24, tube=(24, 32))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    go = oracle.Geometry(pool, tris, shape, off)
    # the same scene with interpolated shading normals on every shape (ffx_smooth): its own pair of geometries
    gds = ops.DeviceGeometry(pool, tris, shape, off, smooth=[True] * len(off))
    gos = oracle.Geometry(pool, tris, shape, off, smooth=[True] * len(off))
    worst_s, bad_s = 0.0, 0
    cam = scene_desc.camera_from_sensor(sc.camera)
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    tex = torch.rand((64, 64, 1), device="cuda")
    tot = flips = lost = 0
    worst = 0.0
    bad_px = 0
    npx = 0
    # the same poses rendered with random principled material rows (include/ffx.h FFX_MAT_*; a fresh table per pose)
    sdm = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=scenes.MAT_STRIDE)
    worst_m, bad_m = 0.0, 0
    for i in range(n_pose):
        a = rng.uniform(-0.2, 0.2)
        R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
        S = np.diag([rng.uniform(0.6, 1.6), rng.uniform(0.8, 1.2), 1.0, 1.0])
        T = np.eye(4)
        T[:3, 3] = rng.uniform(-0.08, 0.08, 3)
        xf = np.stack([np.eye(4)] * (len(off) - 1) + [T @ R @ S]).astype(np.float32)
        offs = off.copy()
        if len(off) > 1:
            offs[1] = off[1] + int(rng.integers(0, nfr[1])) * stride[1]
        gd.update(xf, offs)
        go.update(xf, offs)
        spp = int(rng.choice([1, 4, 64]))
        td, sd_, pd = gd.trace_primary(cam, spp, 1, seed=i)
        to, so, po = go.trace_primary(cam, spp, 1, seed=i)
        pd, sd_ = pd.cpu().numpy(), sd_.cpu().numpy()
        tot += pd.size
        flips += int(((pd != po) | (sd_ != so)).sum())
        lost += int(((pd < 0) & (po >= 0)).sum())
        img_d = gd.render_fwd(sd, torch.from_numpy(alb).cuda(), tex, 16, seed=i).cpu().numpy()
        img_o = go.render_fwd(sd, alb, tex.cpu().numpy(), 16, seed=i)
        err = np.abs(img_d - img_o) / max(float(img_o.max()), 1e-6)
        worst = max(worst, float(err.max()))
        bad_px += int((err > 1e-4).sum())
        npx += err.size
        S_ = len(off)
        mats = np.zeros((S_, scenes.MAT_STRIDE), np.float32)
        mats[:, 0:3] = rng.uniform(0.1, 0.9, (S_, 3))
        mats[:, 3] = 1.0
        for col, (lo, hi) in {4: (0.05, 1.0), 5: (0.0, 1.0), 6: (0.0, 0.5), 7: (0.0, 0.4), 9: (0.0, 1.0), 10: (0.0, 0.5), 11: (0.0, 1.0), 12: (0.0, 1.0),
                              13: (0.0, 1.0), 14: (0.0, 1.0)}.items():
            mats[:, col] = rng.uniform(lo, hi, S_)
        mats[:, 8] = [scenes.specular_to_eta(v) for v in rng.uniform(0.0, 1.0, S_)]
        im_d = gd.render_fwd(sdm, torch.from_numpy(mats).cuda(), tex, 16, seed=i).cpu().numpy()
        im_o = go.render_fwd(sdm, mats, tex.cpu().numpy(), 16, seed=i)
        em = np.abs(im_d - im_o) / max(float(im_o.max()), 1e-6)
        worst_m = max(worst_m, float(em.max()))
        bad_m += int((em > 2e-4).sum())
        if i % 2 == 0:  # every other pose also with interpolated normals (same material table)
            gds.update(xf, offs)
            gos.update(xf, offs)
            is_d = gds.render_fwd(sdm, torch.from_numpy(mats).cuda(), tex, 16, seed=i).cpu().numpy()
            is_o = gos.render_fwd(sdm, mats, tex.cpu().numpy(), 16, seed=i)
            es = np.abs(is_d - is_o) / max(float(is_o.max()), 1e-6)
            worst_s = max(worst_s, float(es.max()))
            bad_s += int((es > 2e-4).sum())
    print(f"poses {n_pose}: rays {tot}, different primitive {flips} ({flips / tot:.2e}), oracle-hit-but-GPU-miss {lost}")
    print(f"render: pixels*channels {npx}, |diff| > 1e-4 of scale: {bad_px} ({bad_px / npx:.2e}), worst {worst:.3e} of scale")
    print(f"render with random principled material rows: |diff| > 2e-4 of scale: {bad_m} ({bad_m / npx:.2e}), worst {worst_m:.3e} of scale")
    print(f"the same with interpolated shading normals (every other pose): |diff| > 2e-4 of scale: {bad_s} ({bad_s / max(npx // 2, 1):.2e}), worst {worst_s:.3e} of scale")


if __name__ == "__main__":
    main()
