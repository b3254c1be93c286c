"""The oracle's Mitsuba half (K7 / K8 / K9, parity-unpinned against Mitsuba itself) against an independent
float64 brute-force restatement (tests/ref_bruteforce.py: no BVH, textbook Moller-Trumbore, written from
SURVEY App. A / DESIGN §4 and not from the oracle's source).  Removes the "same author, same mistake" hole between
the oracle and the HIP kernels; it cannot replace Mitsuba."""
import numpy as np
import pytest

from fireflies_amd import scene_desc, scenes
from tests import ref_bruteforce as bf


def _world(sc, frame=0, xforms=None):
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    S = len(sc.meshes)
    xf = np.tile(np.eye(4, dtype=np.float32), (S, 1, 1)) if xforms is None else xforms
    offs = (off + np.minimum(frame, nfr - 1) * stride).astype(np.int32)
    verts = pool.astype(np.float64).copy()
    gidx = tris + offs[shape][:, None]
    for s in range(S):
        used = np.unique(gidx[shape == s])
        verts[used] = pool[used].astype(np.float64) @ xf[s][:3, :3].T.astype(np.float64) + xf[s][:3, 3]
    return pool, tris, shape, off, offs, xf, alb, verts, gidx


def _xf(S, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(S):
        a = rng.uniform(-0.15, 0.15)
        R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
        T = np.eye(4)
        T[:3, 3] = rng.uniform(-0.05, 0.05, 3)
        out.append(T @ R @ np.diag([rng.uniform(0.8, 1.2), 1.0, 1.0, 1.0]))
    return np.asarray(out, np.float32)


@pytest.mark.parametrize("which", ["hello_world", "vocalfold"])
def test_oracle_agrees_with_an_independent_bruteforce_restatement(oracle, which):
    if which == "hello_world":
        sc, spp, xf, frame = scenes.hello_world(48, 40), 2, None, 0
    else:
        sc, spp, frame = scenes.vocalfold(width=24, height=24, tex=32, frames=3, n_fold=12, tube=(12, 16)), 4, 1
        xf = _xf(len(sc.meshes), 3)
    pool, tris, shape, off, offs, xf, alb, verts, gidx = _world(sc, frame, xf)
    go = oracle.Geometry(pool, tris, shape, off)
    go.update(xf, offs)
    cam = scene_desc.camera_from_sensor(sc.camera)
    # ---- K7: depth and ids, un-jittered and jittered
    for jit in (0, 1):
        t_o, s_o, p_o = go.trace_primary(cam, spp, jit, seed=5)
        t_b, s_b, p_b = bf.trace_primary(verts, gidx, shape, cam, spp, bool(jit), 5)
        same = (p_o == p_b) & (s_o == s_b)
        assert same.mean() >= 0.998, f"{which} jitter={jit}: {1 - same.mean():.2e} of the rays hit a different primitive"
        assert (p_o >= 0).mean() > 0.3
        np.testing.assert_allclose(t_o[same], t_b[same], rtol=2e-5, atol=2e-6)
    # ---- K8: radiance
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    rng = np.random.default_rng(1)
    if sc.projector is not None:
        tex = rng.random((sc.projector.height, sc.projector.width)).astype(np.float32)
    else:
        tex = np.zeros((1, 1), np.float32)
    img_o = go.render_fwd(sd, alb, tex, spp, seed=9)
    img_b = bf.render_fwd(verts, gidx, shape, sd, alb, tex, spp, 9)
    scale = float(img_b.max())
    assert scale > 0.01
    err = np.abs(img_o - img_b)
    # fp32 vs fp64: a sample on an edge (hit or shadow test) can flip: allow 1 % of the pixels an outlier of <= 1.5 / spp
    assert (err > 2e-4 * scale).mean() <= 1e-2, f"{which}: {(err > 2e-4 * scale).mean():.3f} of the pixel channels differ"
    assert err.max() <= 1.5 * scale / spp
    assert abs(float(img_o.mean()) - float(img_b.mean())) <= 2e-3 * float(img_b.mean())
    # without shadows too (the shadow test is the part most sensitive to the lifted origin)
    sd0 = scene_desc.scene_desc(sc, tex_channels=1, shadows=False)
    e0 = np.abs(go.render_fwd(sd0, alb, tex, spp, seed=9) - bf.render_fwd(verts, gidx, shape, sd0, alb, tex, spp, 9))
    assert (e0 > 2e-4 * scale).mean() <= 1e-2 and e0.max() <= 1.5 * scale / spp
    # ---- K9: texture gradient
    if sc.projector is not None:
        gimg = rng.standard_normal((sc.camera.height, sc.camera.width, 3)).astype(np.float32)
        gt_o = go.render_bwd(sd, alb, spp, 9, gimg)[..., 0]
        gt_b = bf.render_bwd(verts, gidx, shape, sd, alb, spp, 9, gimg)
        gs = float(np.abs(gt_b).max())
        assert gs > 0
        gerr = np.abs(gt_o - gt_b)
        assert (gerr > 1e-3 * gs).mean() <= 1e-2 and gerr.max() <= 0.5 * gs
        # the adjoint identity ties K8 and K9 together in the brute-force version itself
        base = bf.render_fwd(verts, gidx, shape, sd, alb, np.zeros_like(tex), spp, 9)
        lhs = float(((img_b - base) * gimg).sum())
        rhs = float((tex.astype(np.float64) * gt_b).sum())
        assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1e-12)
