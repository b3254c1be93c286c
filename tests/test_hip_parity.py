"""GPU parity tests: every HIP entry point of libffx_hip.so is called through the C ABI and
compared with (a) the golden vectors captured from the reference's torch code and (b) the CPU
oracle on the same seeded inputs.  Tolerances are stated per test.  Run with `-m gpu`."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

from fireflies_amd import ops, scenes, scene_desc
from tests.conftest import assert_image_close, load_golden

pytestmark = pytest.mark.gpu

FLIP_Y = np.diag([1.0, -1.0, 1.0, 1.0]).astype(np.float32)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need a HIP device"


# ------------------------------------------------------------------ K1
def test_k1_projection(oracle):
    g = load_golden("g2_projection.npz")
    KF = g["K"] @ FLIP_Y
    for n in (8, 18):
        rays = dev(g[f"rays_{n}"])
        out = host(ops.project_rays_fwd(rays, KF))
        np.testing.assert_allclose(out, g[f"ndc_{n}"], rtol=2e-6, atol=2e-7)  # vs reference
        np.testing.assert_array_equal(out, oracle.project_rays_fwd(g[f"rays_{n}"], KF))  # vs oracle: bit-exact
        gr = host(ops.project_rays_bwd(rays, KF, dev(g[f"gw_{n}"])))
        np.testing.assert_allclose(gr, g[f"grays_{n}"], rtol=2e-5, atol=2e-6)
    g6 = load_golden("g6_math.npz")
    np.testing.assert_allclose(host(ops.transform_points(dev(g6["pts"]), g6["T"], 0)), g6["transform_points"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(ops.transform_points(dev(g6["pts"]), g6["T"], 1)), g6["transform_directions"], rtol=2e-6, atol=1e-6)


# ------------------------------------------------------------------ K2
CASES = ["a", "b", "c", "d", "e", "f", "g"]


@pytest.mark.parametrize("name", CASES)
def test_k2_dense(oracle, name):
    g = load_golden("g3_rasterize_points.npz")
    pts, (s0, s1), sigma = g[f"{name}_pts"], [int(v) for v in g[f"{name}_size"]], float(g[f"{name}_sigma"])
    dense = host(ops.splat_dense_fwd(dev(pts), sigma, s0, s1))
    # values in [0,1]; expf differs by a few ulp between libm / torch / ocml
    np.testing.assert_allclose(dense, g[f"{name}_dense"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(dense, oracle.splat_dense_fwd(pts, sigma, s0, s1), rtol=0, atol=3e-7)
    gp = host(ops.splat_dense_bwd(dev(pts), sigma, s0, s1, dev(g[f"{name}_dense_w"])))
    ref = g[f"{name}_dense_gpts"]
    np.testing.assert_allclose(gp, ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["sum", "softor"])
def test_k2_fused(oracle, name, mode):
    g = load_golden("g3_rasterize_points.npz")
    pts, (s0, s1), sigma = g[f"{name}_pts"], [int(v) for v in g[f"{name}_size"]], float(g[f"{name}_sigma"])
    tex = ops.splat_fwd(dev(pts), sigma, mode, -1, s0, s1)
    np.testing.assert_allclose(host(tex), g[f"{name}_{mode}"], rtol=2e-6, atol=1e-6)
    gp = host(ops.splat_bwd(dev(pts), sigma, mode, -1, s0, s1, tex, dev(g[f"{name}_w"])))
    refg = g[f"{name}_{mode}_gpts"]
    np.testing.assert_allclose(gp, refg, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(refg).max()))
    og = oracle.splat_bwd(pts, sigma, 0 if mode == "sum" else 1, -1, s0, s1, host(tex), g[f"{name}_w"])
    np.testing.assert_allclose(gp, og, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(og).max()))


def test_k2_full_size_and_anchors(oracle):
    g = load_golden("g3_full_500.npz")
    for mode in ("sum", "softor"):
        tex = host(ops.splat_fwd(dev(g["pts"]), 10.0, mode, -1, 500, 500))
        np.testing.assert_allclose(tex, g[mode], rtol=2e-6, atol=1e-6)
    g3 = load_golden("g3_rasterize_points.npz")
    a = host(ops.splat_dense_fwd(dev(np.array([[0.25, 0.75]], np.float32)), 4.0, 8, 16))
    assert a.shape == (1, 16, 8) and np.unravel_index(a.argmax(), a.shape) == (0, 12, 2)
    np.testing.assert_allclose(a, g3["anchor1"], atol=3e-7)


def test_k2_many_points_chunking(oracle):
    # > CAND_MAX (512) candidates per tile and > NEIGH_MAX (2048) points exercise the chunked paths
    rng = np.random.default_rng(3)
    pts = (rng.random((2500, 2)) * 0.2 + 0.4).astype(np.float32)
    for mode, red in (("sum", 0), ("softor", 1)):
        tex = ops.splat_fwd(dev(pts), 10.0, mode, -1, 64, 48)
        np.testing.assert_allclose(host(tex), oracle.splat_fwd(pts, 10.0, red, -1, 64, 48), rtol=3e-6, atol=1e-6)
    w = rng.standard_normal((48, 64)).astype(np.float32)
    sub = pts[:2100]
    tex = ops.splat_fwd(dev(sub), 10.0, "sum", -1, 64, 48)
    gp = host(ops.splat_bwd(dev(sub), 10.0, "sum", -1, 64, 48, tex, dev(w)))
    og = oracle.splat_bwd(sub, 10.0, 0, -1, 64, 48, host(tex), w)
    np.testing.assert_allclose(gp, og, rtol=1e-3, atol=1e-4 * np.abs(og).max())


def test_k2_empty_and_outside(oracle):
    z = ops.splat_fwd(torch.empty((0, 2), device="cuda"), 10.0, "sum", -1, 20, 10)
    assert z.shape == (10, 20) and float(z.abs().sum()) == 0.0
    z = ops.splat_fwd(torch.empty((0, 2), device="cuda"), 10.0, "softor", -1, 20, 10)
    assert float(z.abs().sum()) == 0.0
    far = dev(np.array([[5.0, 5.0], [-3.0, 0.5]], np.float32))
    assert float(ops.splat_fwd(far, 10.0, "sum", -1, 32, 32).abs().sum()) == 0.0
    assert float(ops.splat_fwd(far, 100.0, "sum", 20, 32, 32).abs().sum()) == 0.0
    gp = ops.splat_bwd(far, 10.0, "sum", -1, 32, 32, None, torch.ones(32, 32, device="cuda"))
    assert float(gp.abs().sum()) == 0.0


def _baked_sum_grad_f64(pts, w, S, sig, half):
    # positions are the fp32 products p*size, exactly as the reference and the kernels form them
    scaled = (pts.astype(np.float32) * np.float32(S)).astype(np.float64)
    w = w.astype(np.float64)
    gp = np.zeros(pts.shape, np.float64)
    for k, (p0, p1) in enumerate(scaled):
        f0, f1 = int(np.floor(p0)), int(np.floor(p1))
        A = np.arange(f0 - half, f0 + half + 1)
        B = np.arange(f1 - half, f1 + half + 1)
        yd, xd = np.meshgrid(A - p0, B - p1, indexing="xy")
        d = yd * yd + xd * xd
        c = np.exp(-((d / sig) ** 2)) * 4 * d / sig**2 * w[np.ix_(B, A)]
        gp[k] = [(c * yd).sum() * S, (c * xd).sum() * S]
    return gp


@pytest.mark.parametrize("tag", ["in", "bd"])
def test_k2_baked(oracle, tag):
    g = load_golden("g4_baked.npz")
    pts = dev(g[f"{tag}_pts"])
    out = host(ops.splat_fwd(pts, 100.0, "sum", 20, 100, 100))
    np.testing.assert_allclose(out, g[f"{tag}_baked_sum"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out.T, g[f"{tag}_baked_sum_2"], rtol=2e-6, atol=1e-6)
    out = host(ops.splat_fwd(pts, 100.0, "softor", 25, 100, 100))
    np.testing.assert_allclose(out, g[f"{tag}_baked_softor"], rtol=2e-6, atol=1e-6)
    s0, s1 = [int(v) for v in g["ns_size"]]
    out = host(ops.splat_fwd(dev(g["ns_pts"]), 16.0, "sum", 8, s0, s1))
    np.testing.assert_allclose(out, g["ns_baked_sum"], rtol=2e-6, atol=1e-6)
    tex = ops.splat_fwd(dev(g["in_pts"]), 100.0, "sum", 20, 100, 100)
    gp = host(ops.splat_bwd(dev(g["in_pts"]), 100.0, "sum", 20, 100, 100, tex, dev(g["in_baked_sum_w"])))
    # heavily cancelling sum (|terms| ~ 1e-2, result ~ 3e-4; see test_oracle_golden): judge both the
    # kernel and the reference's own fp32 autograd against a float64 evaluation of the same formula
    f64 = _baked_sum_grad_f64(g["in_pts"], g["in_baked_sum_w"], 100, 100.0, 20)
    # noise floor: 1681 fp32 terms of magnitude <= 5e-2 (rounding ~1e-8 each), summed, times size0 = 100
    # -> ~4e-5 for ANY fp32-term implementation (oracle: 2.2e-5, reference autograd: 3.5e-5)
    assert np.abs(gp - f64).max() <= 8e-5
    assert np.abs(g["in_baked_sum_gpts"] - f64).max() <= 8e-5
    # softor gradient, baked window, vs oracle
    w = np.cos(np.arange(10000, dtype=np.float32) * 0.13).reshape(100, 100)
    tex = ops.splat_fwd(dev(g["bd_pts"]), 100.0, "softor", 25, 100, 100)
    gp = host(ops.splat_bwd(dev(g["bd_pts"]), 100.0, "softor", 25, 100, 100, tex, dev(w)))
    og = oracle.splat_bwd(g["bd_pts"], 100.0, 1, 25, 100, 100, host(tex), w)
    np.testing.assert_allclose(gp, og, rtol=1e-3, atol=5e-5)


def test_k2_depth_and_lines(oracle):
    g = load_golden("g5_depth_lines.npz")
    s0, s1 = [int(v) for v in g["depth_size"]]
    out = host(ops.splat_depth_fwd(dev(g["depth_pts"][:, :2]), dev(g["depth_pts"][:, 2]), 6.0, s0, s1))
    np.testing.assert_allclose(out, g["depth_out"], rtol=2e-6, atol=1e-6)
    l0, l1 = [int(v) for v in g["lines_size"]]
    out = host(ops.splat_lines_fwd(dev(g["lines_in"]), 3.0, l0, l1))
    np.testing.assert_allclose(out, g["lines_out"], rtol=1e-5, atol=1e-6)
    # gradient w.r.t. the segments: the reference's autograd (golden g10) and the oracle
    g10 = load_golden("g10_lines_grad.npz")
    for t in "abc":
        s0, s1 = [int(v) for v in g10[f"{t}_size"]]
        sigma = float(g10[f"{t}_sigma"])
        gl = host(ops.splat_lines_bwd(dev(g10[f"{t}_lines"]), sigma, s0, s1, dev(g10[f"{t}_w"])))
        ref = g10[f"{t}_glines"]
        np.testing.assert_allclose(gl, ref, rtol=1e-4, atol=3e-6 * np.abs(ref).max())
        np.testing.assert_allclose(gl, oracle.splat_lines_bwd(g10[f"{t}_lines"], sigma, s0, s1, g10[f"{t}_w"]), rtol=2e-5, atol=1e-6 * np.abs(ref).max())
    rng = np.random.default_rng(3)  # a non-square film and many segments
    lines = (rng.random((37, 2, 2)) * 1.2 - 0.1).astype(np.float32)
    w = rng.standard_normal((37, 40, 56)).astype(np.float32)
    gl = host(ops.splat_lines_bwd(dev(lines), 25.0, 56, 40, dev(w)))
    ref = oracle.splat_lines_bwd(lines, 25.0, 56, 40, w)
    np.testing.assert_allclose(gl, ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max())


# ------------------------------------------------------------------ K3
def test_renders_below_64_spp_pack_several_pixels_into_a_wave(oracle, monkeypatch):
    """k_render_fwd_blk (round 5): below 33 samples per pixel ffx_render_fwd gives a wave a compact block of pixels instead of one (K7's layout):
    sample counts from 1 to 32 — powers of two and not — on a film whose sides are no multiple of any block, fp32 and fp16, Lambert and material
    rows, with and without shadow rays and tile bins: against the oracle within the radiance bounds, and bit for bit the image of the
    pixel-per-wave kernel (FFX_RENDER_BLOCKS=0: the same samples, the same summation tree)."""
    for k in ("FFX_TRAVERSAL", "FFX_WIDE", "FFX_BINS", "FFX_RENDER_BLOCKS"):
        monkeypatch.delenv(k, raising=False)
    for principled in (False, True):
        sc = scenes.vocalfold(width=45, height=37, tex=64, frames=3, n_fold=20, tube=(20, 24), principled=principled)
        go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 8))
        tex = _tex(sc, 1)
        for shadows in (True, False):
            sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=shadows)
            for spp in ((1, 2, 3, 4, 7, 8, 16, 24, 32) if shadows else (1, 5, 32)):
                want = go.render_fwd(sd, alb, host(tex), spp, seed=21)
                for env in ({}, {"FFX_BINS": "0"}, {"FFX_WIDE": "0", "FFX_BINS": "0"}):
                    for k, v in env.items():
                        monkeypatch.setenv(k, v)
                    got = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=21))
                    monkeypatch.setenv("FFX_RENDER_BLOCKS", "0")
                    one = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=21))
                    monkeypatch.delenv("FFX_RENDER_BLOCKS")
                    for k in env:
                        monkeypatch.delenv(k)
                    scale, _ = _assert_image_close(got, want, spp, frac=1e-3, what=f"pixel blocks, {spp} spp, {env}, principled {principled}")
                    assert scale > 0.02
                    np.testing.assert_array_equal(got.view(np.uint32), one.view(np.uint32), err_msg=f"{spp} spp {env}")  # (the same summation tree)
                if spp in (1, 7, 32):
                    h16 = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=21, fp16=True).float())
                    np.testing.assert_allclose(h16, got, rtol=2e-3, atol=1e-4 * scale)


def test_filtered_renders_below_33_spp_pack_several_pixels_into_a_wave(oracle, monkeypatch):
    """k_render_fwd_blk<..., RF> (round 5): the gaussian film — the film of every scene the reference loads, whose examples render at 10 and 12 spp
    (examples/01_hello_world.py:29, 06_animation.py:51) — below 33 samples per pixel: 8 / 4 / 2 pixels per wave whose 25 x 4 outgoing sums are formed by
    rf_fold_blk; against the oracle, and bit for bit the image of the pixel-per-wave kernel (FFX_RENDER_BLOCKS=0: the same fma chains), on a film whose
    sides are no multiple of a block, Lambert and material rows, 1- and 3-channel textures, fp16."""
    from fireflies_amd import _abi
    from tests.test_bruteforce_cpu import material_rows

    monkeypatch.delenv("FFX_RENDER_BLOCKS", raising=False)
    sc = scenes.vocalfold(width=53, height=42, tex=96, frames=3, n_fold=24, tube=(24, 32))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 2))
    for ch, rows, stddev in ((1, alb, 0.5), (3, alb, 0.5), (1, material_rows(len(sc.meshes), 5), 0.5), (1, alb, 0.3)):
        sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=_abi.MAT_STRIDE if rows.shape[1] == _abi.MAT_STRIDE else 0, rfilter=("gaussian", stddev))
        tex = _tex(sc, ch)
        for spp in (1, 3, 8, 10, 12, 16, 27, 32):
            got = gd.render_fwd(sd, dev(rows), tex, spp, seed=11)
            monkeypatch.setenv("FFX_RENDER_BLOCKS", "0")
            one = gd.render_fwd(sd, dev(rows), tex, spp, seed=11)
            monkeypatch.delenv("FFX_RENDER_BLOCKS")
            assert torch.equal(got.view(torch.int32), one.view(torch.int32)), (ch, rows.shape, spp)
            if spp in (1, 10, 32):
                want = go.render_fwd(sd, rows, host(tex), spp, seed=11)
                scale, _ = _assert_image_close(host(got), want, spp, frac=0.02, what=f"filtered pixel blocks ch={ch} spp={spp}")
                h16 = host(gd.render_fwd(sd, dev(rows), tex, spp, seed=11, fp16=True)).astype(np.float32)
                np.testing.assert_allclose(h16, host(got), rtol=1e-3, atol=1e-4 * scale)


def test_dataset_path_steps_in_single_launches(oracle):
    """include/ffx.h ffx_silhouette_fwd / ffx_noise_clamp / ffx_rgb_to_gray (ABI 8): each against the oracle's long form (mask -> its K3
    blur -> product; the clipped sum; the weighted sum) and against the torch expressions they replace in fireflies_amd.postprocessing — bit for bit,
    which is what lets the dataset loop swap ~20 small launches per sample for three; discs that leave the image, reflect borders, ragged tile edges,
    NaN kept by the clamp, the fp16 film."""
    from fireflies_amd import _abi

    olib = oracle.api().lib
    for name in ("ffx_silhouette_fwd", "ffx_noise_clamp", "ffx_rgb_to_gray"):
        getattr(olib, name).restype, getattr(olib, name).argtypes = _abi.PROTOTYPES[name]
    g = torch.Generator(device="cpu").manual_seed(3)
    for (h, w, cx, cy, r, ks, sg) in ((512, 512, 150, 250, 200, 11, 5.0), (37, 53, 5, 30, 12, 11, 5.0), (64, 40, -20, 10, 45, 7, 2.0), (9, 70, 60, 4, 0, 15, 4.0),
                                      (33, 33, 16, 16, 400, 3, 1.0)):
        img = torch.rand((h, w), generator=g)
        d = img.to("cuda")
        got = ops.silhouette(d, cx, cy, r, ks, sg)
        want = np.empty((h, w), np.float32)
        assert olib.ffx_silhouette_fwd(img.numpy().ctypes.data, h, w, cx, cy, r, ks, sg, want.ctypes.data, None) == 0
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=str((h, w, cx, cy, r)))
        yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
        mask = (((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).float().contiguous()
        assert torch.equal(got, d * ops.blur_fwd(mask, ks, sg))  # (the torch form it replaces)
        assert torch.equal(ops.silhouette(d, cx, cy, r, ks, sg), got)
    with pytest.raises(Exception):
        ops.silhouette(torch.rand((8, 8), device="cuda"), 1, 1, 2, 4, 1.0)  # even kernel size
    for n, mean, sd in ((512 * 512, 0.0, 0.05), (1000, 0.1, 0.02), (77, -0.05, 0.2)):
        img, nz = torch.rand(n, generator=g), torch.randn(n, generator=g)
        img[3], nz[5] = float("nan"), float("inf")
        want = np.empty(n, np.float32)
        assert olib.ffx_noise_clamp(img.numpy().ctypes.data, nz.numpy().ctypes.data, n, mean, sd, 0.0, 1.0, want.ctypes.data, None) == 0
        di, dn = img.to("cuda"), nz.to("cuda")
        torch_form = torch.clamp(di + (dn * sd + mean), 0, 1)
        got = ops.noise_clamp(di, dn.clone(), mean, sd)
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert torch.equal(got.view(torch.int32), torch_form.view(torch.int32)) and bool(torch.isnan(got[3])) and float(got[5]) == 1.0
    for n in (512 * 512, 129):
        rgb = torch.rand((n, 3), generator=g) * 3.0
        for dt in (torch.float32, torch.float16):
            src = rgb.to(dt)
            want = np.empty(n, np.float32)
            assert olib.ffx_rgb_to_gray(src.numpy().ctypes.data, int(dt == torch.float16), n, 0.299, 0.587, 0.114, want.ctypes.data, None) == 0
            dsrc = src.to("cuda")
            got = ops.rgb_to_gray(dsrc)
            np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
            f = dsrc.float()
            assert torch.equal(got, f[..., 0] * 0.299 + f[..., 1] * 0.587 + f[..., 2] * 0.114)
    # the relabelling of depth.py:119-125 in one reduction (graphics.depth._labels): ids -= min; ids = max - ids
    from fireflies_amd.graphics import depth

    ids = torch.randint(-1, 5, (4096,), generator=g).to(torch.int32).to("cuda")
    ptr = ids.to(torch.int64) + 1
    ptr = ptr - ptr.min()
    assert torch.equal(depth._labels(ids), ptr.max() - ptr)
    assert torch.equal(depth._labels(torch.full((7,), -1, dtype=torch.int32, device="cuda")), torch.zeros(7, dtype=torch.int64, device="cuda"))


def test_k3_blur(oracle):
    rng = np.random.default_rng(0)
    for shape in ((500, 500), (37, 61), (6, 7), (3, 4)):
        a = rng.random(shape).astype(np.float32)
        np.testing.assert_array_equal(host(ops.blur_fwd(dev(a))), oracle.blur_fwd(a))  # same fma order: bit-exact
        g = rng.standard_normal(shape).astype(np.float32)
        np.testing.assert_allclose(host(ops.blur_bwd(dev(g))), oracle.blur_bwd(g), rtol=1e-5, atol=1e-6)
    a = rng.random((40, 33)).astype(np.float32)
    np.testing.assert_array_equal(host(ops.blur_fwd(dev(a), 9, 2.0)), oracle.blur_fwd(a, 9, 2.0))


_assert_image_close = assert_image_close  # (tests/conftest.py)


# ------------------------------------------------------------------ K5..K7
def _pair(oracle, sc, frame=0, xforms=None):
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    S = len(sc.meshes)
    if xforms is None:
        xforms = np.tile(np.eye(4, dtype=np.float32), (S, 1, 1))
    offs = (off + np.minimum(frame, nfr - 1) * stride).astype(np.int32)
    go = oracle.Geometry(pool, tris, shape, off)
    go.update(xforms, offs)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    gd.update(xforms, offs)
    return go, gd, alb


def _rand_xforms(S, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(S):
        a = rng.uniform(-0.15, 0.15)
        R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
        Sx = np.diag([rng.uniform(0.8, 1.2), 1.0, 1.0, 1.0])
        T = np.eye(4)
        T[:3, 3] = rng.uniform(-0.05, 0.05, 3)
        out.append(T @ R @ Sx)
    return np.asarray(out, np.float32)


def _cmp_hits(t_d, s_d, p_d, t_o, s_o, p_o, what):
    t_d, s_d, p_d = host(t_d), host(s_d), host(p_d)
    same = (s_d == s_o) & (p_d == p_o)
    frac = 1.0 - same.mean()
    # identical operation order => identical hits; allow 1e-4 of the rays for 1-ulp edge flips
    assert frac <= 1e-4, f"{what}: {frac:.2e} of the rays hit a different primitive"
    np.testing.assert_allclose(t_d[same], t_o[same], rtol=1e-5, atol=1e-6, err_msg=what)
    return float((t_d[same] == t_o[same]).mean())


@pytest.mark.parametrize("cfg", ["hello", "vocalfold_small", "vocalfold_full"])
def test_k7_trace_primary(oracle, cfg):
    if cfg == "hello":
        sc, W, H, spp = scenes.hello_world(64, 48), 64, 48, 2
    elif cfg == "vocalfold_small":
        sc = scenes.vocalfold(width=96, height=80, tex=64, frames=3, n_fold=16, tube=(24, 24))
        W, H, spp = 96, 80, 2
    else:
        sc = scenes.vocalfold(width=128, height=128, frames=4)
        W, H, spp = 128, 128, 2
    xf = _rand_xforms(len(sc.meshes), 1)
    go, gd, _ = _pair(oracle, sc, frame=2, xforms=xf)
    cam = scene_desc.camera_from_sensor(sc.camera)
    for jitter in (0, 1):
        td, sd_, pd = gd.trace_primary(cam, spp, jitter, seed=7)
        to, so, po = go.trace_primary(cam, spp, jitter, seed=7)
        exact = _cmp_hits(td, sd_, pd, to, so, po, f"{cfg} jitter={jitter}")
        assert exact > 0.999, f"only {exact:.4f} of the depths are bit-identical"
        assert (host(sd_) >= 0).mean() > 0.3
    # misses write t = 0, ids -1 (depth.py:84)
    miss = host(sd_) < 0
    assert (host(td)[miss] == 0).all() and (host(pd)[miss] == -1).all()


@pytest.mark.parametrize("spp", [1, 3, 16, 64, 70])
def test_k7_packet_layouts(oracle, spp, monkeypatch):
    """Every wave layout of the packet K7 kernel (8x8 pixels x 1 sample ... 1 pixel x 64 samples,
    a sample count that does not divide 64, more than one pass) and the per-lane kernel give the
    oracle's hits at an image size that is not a multiple of the pixel block."""
    sc = scenes.vocalfold(width=52, height=37, tex=64, frames=3, n_fold=16, tube=(24, 24))
    go, gd, _ = _pair(oracle, sc, frame=1, xforms=_rand_xforms(len(sc.meshes), 4))
    cam = scene_desc.camera_from_sensor(sc.camera)
    to, so, po = go.trace_primary(cam, spp, 1, seed=3)
    for mode, bins in (("packet", "1"), ("packet", "0"), ("lane", "1")):  # (FFX_BINS=0: the packet tree walks that the tile bins fall back to)
        monkeypatch.setenv("FFX_TRAVERSAL", mode)
        monkeypatch.setenv("FFX_BINS", bins)
        td, sd_, pd = gd.trace_primary(cam, spp, 1, seed=3)
        assert td.shape[0] == 52 * 37 * spp
        _cmp_hits(td, sd_, pd, to, so, po, f"spp={spp} {mode} bins={bins}")


def test_k7_trace_rays_laser(oracle):
    sc = scenes.vocalfold(width=64, height=64, frames=2, n_fold=32, tube=(32, 32))
    go, gd, _ = _pair(oracle, sc, frame=1)
    rng = np.random.default_rng(5)
    n = 1000
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 1.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.array([[0.25, 0.0, 1.5]], np.float32), (n, 1))
    td, sd_, pd = gd.trace_rays(dev(o), dev(d))
    to, so, po = go.trace_rays(o, d)
    _cmp_hits(td, sd_, pd, to, so, po, "laser rays")
    z = gd.trace_rays(torch.empty((0, 3), device="cuda"), torch.empty((0, 3), device="cuda"))
    assert z[0].numel() == 0


def test_k5k6_update_is_idempotent_and_tracks_frames(oracle):
    sc = scenes.vocalfold(width=64, height=64, frames=6, n_fold=24, tube=(24, 32))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    go = oracle.Geometry(pool, tris, shape, off)
    cam = scene_desc.camera_from_sensor(sc.camera)
    xf = _rand_xforms(2, 9)
    ref = None
    for frame in (3, 0, 5, 3):
        offs = off + np.array([0, frame * stride[1]], np.int32)
        gd.update(xf, offs)
        go.update(xf, offs)
        td, sd_, pd = gd.trace_primary(cam, 1, 0, 0)
        to, so, po = go.trace_primary(cam, 1, 0, 0)
        _cmp_hits(td, sd_, pd, to, so, po, f"frame {frame}")
        if frame == 3:
            if ref is None:
                ref = host(td).copy()
            else:
                np.testing.assert_array_equal(host(td), ref)  # same pose -> same bits after other poses
    # refit is a pure function of (pose, topology): nodes + records are identical after a repeated update
    # (which lands in the other of the two blobs; the apex areas behind the records are per-call scratch)
    geo = int(gd.info.off_recs) + 48 * int(gd.info.n_tris)
    blob1 = gd.blob[:geo].clone()
    gd.update(xf, offs)
    assert torch.equal(blob1, gd.blob[:geo])


@pytest.mark.parametrize("cfg", ["hello", "one_leaf", "small", "vocalfold", "colon", "colon_small_treelets"])
def test_k5k6_fused_update_writes_the_same_blob_as_the_level_launches(cfg, monkeypatch):
    """ffx_scene_update by treelets: a workgroup (round 4: one wave) per treelet writes its records, per-triangle boxes, its nodes height
    by height behind workgroup barriers and the wide children that live in its nodes; the top of the tree follows as a second, one-workgroup
    launch (round 4, the default) or is done by the workgroup that arrives last in the same launch (round 3, FFX_REFIT=fused).  Both must
    write, bit for bit, what the eight dependent launches they replace write (FFX_REFIT=levels: records, level by level, tail, wide boxes):
    nodes, records, wide nodes, triangle boxes — for a scene that is a single leaf, a tree below one treelet, the vocal fold (~150 treelets of
    <= 512 triangles) and the colon (~1500; `colon_small_treelets`, misnamed since round 4, runs the 4096-triangle cut: ~190 treelets), over
    repeated updates of the same blob (the fused form's arrival counter must come back to zero)."""
    if cfg == "hello":
        sc = scenes.hello_world(32, 32)
    elif cfg == "one_leaf":
        sc = scenes.hello_world(32, 32)
        sc.meshes = [scenes.MeshData("mesh-Tri", sc.meshes[0].frames[:, :3].copy(), np.array([[0, 1, 2]], np.int32), (0.5, 0.5, 0.5), "mat")]
    elif cfg == "small":
        sc = scenes.vocalfold(width=32, height=32, tex=32, frames=3, n_fold=20, tube=(20, 24))
    elif cfg == "vocalfold":
        sc = scenes.vocalfold(width=32, height=32, tex=32, frames=4)
    else:
        sc = scenes.colon(width=32, height=32, tex=32)
    monkeypatch.delenv("FFX_TREELET_TRIS", raising=False)
    if cfg == "colon_small_treelets":
        monkeypatch.setenv("FFX_TREELET_TRIS", "4096")  # (read by the host builder)
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    S = len(sc.meshes)
    monkeypatch.setenv("FFX_ASYNC_UPDATE", "0")  # one blob each, everything on the current stream
    monkeypatch.delenv("FFX_REFIT", raising=False)
    gf = ops.DeviceGeometry(pool, tris, shape, off)
    gl = ops.DeviceGeometry(pool, tris, shape, off)
    info = gf.info
    assert info.n_treelets >= 1 and info.off_plan > 0
    if cfg == "colon":
        assert info.n_treelets > 1000
    if cfg == "colon_small_treelets":
        assert 100 <= info.n_treelets <= 256
    end = int(info.off_whdr)  # nodes, order, refit list, records, wide nodes, triangle boxes, wsrc
    for it in range(4):
        xf = _rand_xforms(S, 40 + it)
        offs = (off + np.minimum(it, nfr - 1) * stride).astype(np.int32)
        if it % 2 == 0:
            monkeypatch.delenv("FFX_REFIT", raising=False)  # treelets + top: two launches
        else:
            monkeypatch.setenv("FFX_REFIT", "fused")  # one launch, the last workgroup re-fits the top
        gf.update(xf, offs)
        monkeypatch.setenv("FFX_REFIT", "levels")
        gl.update(xf, offs)
        a, b = gf.blob[:end].clone(), gl.blob[:end].clone()
        if not torch.equal(a, b):
            d = torch.nonzero(a != b).reshape(-1)
            raise AssertionError(f"{cfg} update {it}: {d.numel()} bytes differ, first at {int(d[0])} (nodes at {info.off_nodes}, recs at {info.off_recs}, "
                                 f"wnodes at {info.off_wnodes}, tq at {info.off_tq})")
        plan_tail = gf.blob[int(info.off_plan) + 4 * (int(info.plan_ints) - 1):][:4].view(torch.int32)
        assert int(plan_tail[0]) == 0  # arrival counter reset by the last workgroup
    # the device-table entry point (ffx_scene_update) takes the same path
    monkeypatch.delenv("FFX_REFIT", raising=False)
    gf.update(torch.from_numpy(xf).cuda(), offs)
    assert torch.equal(gf.blob[:end], b)


def test_k5k6_async_double_buffered_update_matches_synchronous(monkeypatch):
    """update() re-fits the blob that is not being read, on a side stream, while earlier traces may still
    be running; a burst of (update, trace) pairs without any host sync must give exactly what the
    single-blob, single-stream path gives for every pose."""
    sc = scenes.vocalfold(width=96, height=96, frames=6, n_fold=24, tube=(24, 32))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    cam = scene_desc.camera_from_sensor(sc.camera)
    poses = []
    rng = np.random.default_rng(4)
    for i in range(12):
        offs = off.copy()
        offs[1] = off[1] + int(rng.integers(0, nfr[1])) * stride[1]
        poses.append((_rand_xforms(2, 100 + i), offs))
    monkeypatch.setenv("FFX_ASYNC_UPDATE", "1")
    ga = ops.DeviceGeometry(pool, tris, shape, off)
    assert ga._async and len(ga._blobs) == 4
    runs = []
    for last_spp, ring in ((64, 2), (8, 4)):  # (the ring of blob copies update() rotates through: two behind renders at 33 spp and more, all four below)
        ga._last_spp = last_spp
        assert ga._ring() == ring
        outs, used = [], set()
        for xf, offs in poses:  # no synchronisation anywhere in this loop
            ga.update(xf, offs)
            used.add(ga._cur)
            outs.append(ga.trace_primary(cam, 4, 1, seed=9))
        assert len(used) == ring
        runs.append(outs)
    monkeypatch.setenv("FFX_ASYNC_UPDATE", "0")
    gs = ops.DeviceGeometry(pool, tris, shape, off)
    assert not gs._async
    for i, (xf, offs) in enumerate(poses):
        gs.update(xf, offs)
        t2, s2, p2 = gs.trace_primary(cam, 4, 1, seed=9)
        for outs in runs:
            t, s_, p = outs[i]
            assert torch.equal(t, t2) and torch.equal(s_, s2) and torch.equal(p, p2)


def test_async_update_with_caller_supplied_vertices(monkeypatch):
    """the animation path (`<mesh>.vertex_positions` assignment, add_animation_func, OBJ sequences): Scene
    copies the caller's vertices into a scratch slot of the vertex pool on the caller's stream while the
    refit runs on a side stream.  A burst of (new vertices, update, trace) without any host sync must see,
    for every pose, exactly the vertices of that pose — i.e. what the single-stream path computes."""
    from fireflies_amd import mi

    def run(async_flag):
        monkeypatch.setenv("FFX_ASYNC_UPDATE", async_flag)
        sc = scenes.vocalfold(width=96, height=96, frames=2, n_fold=24, tube=(24, 32))
        ms = mi.load_scene_data(sc)
        assert ms.geom._async == (async_flag == "1")
        params = mi.traverse(ms)
        cam = ms.camera_struct(0)
        base = torch.from_numpy(sc.meshes[1].frames[0]).to("cuda")
        outs = []
        g = torch.Generator(device="cuda").manual_seed(5)
        for i in range(10):  # no synchronisation anywhere in this loop
            v = base * (1.0 + 0.3 * torch.rand((), device="cuda", generator=g)) + 0.05 * torch.randn(3, device="cuda", generator=g)
            params[sc.meshes[1].name + ".vertex_positions"] = mi.Float32(v.reshape(-1))
            params.update()
            outs.append(ms.geom.trace_primary(cam, 2, 1, seed=i))
        torch.cuda.synchronize()
        return outs

    a, b = run("1"), run("0")
    for (t1, s1, p1), (t2, s2, p2) in zip(a, b):
        assert torch.equal(t1, t2) and torch.equal(s1, s2) and torch.equal(p1, p2)
    assert not torch.equal(a[0][0], a[1][0])  # the poses really differ


def test_l1_value_grad_matches_oracle(oracle):
    rng = np.random.default_rng(6)
    for shape in ((500, 500), (7, 3), (1, 1)):
        a = rng.random(shape).astype(np.float32)
        b = rng.random(shape).astype(np.float32)
        b.flat[0] = a.flat[0]
        v, g = ops.l1_value_grad(dev(a), dev(b), 0.1)
        vo, go = oracle.l1_value_grad(a, b, 0.1)
        assert abs(float(v) - vo) <= 2e-6 * max(vo, 1e-6)
        np.testing.assert_array_equal(host(g), go)
        # ffx_l1_value_grad_acc: the same value also joins a running sum inside the reduction launch (a step's loss over its scene samples)
        acc = torch.full((3,), 2.5, device="cuda")
        v2, g2 = ops.l1_value_grad(dev(a), dev(b), 0.1, acc=acc[1])
        v3, _ = ops.l1_value_grad(dev(a), dev(b), 0.1, acc=acc[1])
        assert float(v2) == float(v) == float(v3) and torch.equal(g2, g)
        assert host(acc).tolist() == [2.5, float(np.float32(np.float32(2.5) + np.float32(float(v))) + np.float32(float(v))), 2.5]


def test_clamp_to_fov_matches_oracle(oracle):
    K = scenes.perspective_projection(64, 64, 30.0, 0.01, 100.0).astype(np.float64)
    KF = (K @ np.diag([1.0, -1.0, 1.0, 1.0])).astype(np.float32)
    KI = np.linalg.inv(KF.astype(np.float64)).astype(np.float32)
    rng = np.random.default_rng(8)
    r = rng.standard_normal((333, 3)).astype(np.float32)
    r[:, 2] = -np.abs(r[:, 2]) - 1.5
    for nn in (1, 2):
        d = dev(r.copy())
        out = host(ops.clamp_to_fov_(d, KF, KI, 0.05, 0.95, nn))
        np.testing.assert_allclose(out, oracle.clamp_to_fov(r, KF, KI, 0.05, 0.95, nn), rtol=0, atol=3e-7)
    assert ops.clamp_to_fov_(torch.empty((0, 3), device="cuda"), KF, KI, 0.05, 0.95).shape == (0, 3)


# ------------------------------------------------------------------ K8 / K9
def _tex(sc, ch=1, seed=0):
    rng = np.random.default_rng(seed)
    n = 64
    pts = (rng.random((n, 2)) * 0.8 + 0.1).astype(np.float32)
    t = ops.blur_fwd(ops.splat_fwd(dev(pts), 10.0, "sum", -1, sc.projector.width, sc.projector.height))
    if ch == 3:
        t = torch.stack([0.2 * t, t, 0.1 * t], -1).contiguous()
    return t


@pytest.mark.parametrize("shadows", [False, True])
@pytest.mark.parametrize("ch", [1, 3])
def test_k8_render_forward(oracle, shadows, ch):
    sc = scenes.vocalfold(width=72, height=64, tex=96, frames=3, n_fold=24, tube=(24, 32))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 2))
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=shadows)
    tex = _tex(sc, ch)
    img_d = host(gd.render_fwd(sd, dev(alb), tex, 8, seed=11))
    img_o = go.render_fwd(sd, alb, host(tex), 8, seed=11)
    assert img_o.max() > 0.05
    scale, _ = _assert_image_close(img_d, img_o, 8, what=f"shadows={shadows} ch={ch}")
    # fp16 film (config 5): converted once at the store
    img_h = host(gd.render_fwd(sd, dev(alb), tex, 8, seed=11, fp16=True)).astype(np.float32)
    np.testing.assert_allclose(img_h, img_d, rtol=1e-3, atol=1e-4 * scale)


def test_k8_hello_world_plumbing(oracle):
    sc = scenes.hello_world(64, 64)
    go, gd, alb = _pair(oracle, sc)
    sd = scene_desc.scene_desc(sc, shadows=True)
    img_d = host(gd.render_fwd(sd, dev(alb), None, 16, seed=0))
    img_o = go.render_fwd(sd, alb, np.zeros((1, 1), np.float32), 16, seed=0)
    scale, _ = _assert_image_close(img_d, img_o, 16, frac=1e-3, what="hello_world")
    assert scale > 0.01


@pytest.mark.parametrize("ch", [1, 3])
def test_k9_render_backward(oracle, ch):
    sc = scenes.vocalfold(width=72, height=64, tex=96, frames=3, n_fold=24, tube=(24, 32))
    go, gd, alb = _pair(oracle, sc, frame=2, xforms=_rand_xforms(2, 4))
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True)
    rng = np.random.default_rng(1)
    gimg = rng.standard_normal((64, 72, 3)).astype(np.float32)
    gt_d = host(gd.render_bwd(sd, dev(alb), 8, 13, dev(gimg)))
    gt_o = go.render_bwd(sd, alb, 8, 13, gimg)
    scale = float(np.abs(gt_o).max())
    assert scale > 0
    # gradient: atomic accumulation order -> 1e-3 of the gradient scale per texel
    err = np.abs(gt_d - gt_o)
    assert (err > 1e-3 * scale).mean() <= 2e-4, f"{(err > 1e-3 * scale).mean():.2e}"
    assert err.max() <= 0.1 * scale


def test_k8k9_full_size_properties():
    """BASELINE size (512x512, 64 spp, 53,248 triangles): size-independent properties."""
    sc = scenes.vocalfold()
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    gd.update(_rand_xforms(2, 6), off + np.array([0, 17 * stride[1]], np.int32))
    sd = scene_desc.scene_desc(sc, shadows=True)
    albd = dev(alb)
    tex = _tex(sc)
    img = gd.render_fwd(sd, albd, tex, 64, seed=3)
    assert torch.isfinite(img).all() and float(img.max()) > 0.05
    # deterministic: no atomics in the forward pass
    assert torch.equal(img, gd.render_fwd(sd, albd, tex, 64, seed=3))
    # linear in the texture
    base = gd.render_fwd(sd, albd, torch.zeros_like(tex), 64, seed=3)
    img3 = gd.render_fwd(sd, albd, 3.0 * tex, 64, seed=3)
    torch.testing.assert_close(img3 - base, 3.0 * (img - base), rtol=1e-4, atol=1e-5 * float(img.max()))
    # forward / adjoint dot-product identity  <J tex, g> = <tex, J^T g>
    g = torch.randn_like(img)
    gtex = gd.render_bwd(sd, albd, 64, 3, g)
    lhs = float(((img - base).double() * g.double()).sum())
    rhs = float((tex.double() * gtex[..., 0].double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    # different seeds give different noise but the same mean to Monte-Carlo accuracy
    img_b = gd.render_fwd(sd, albd, tex, 64, seed=4)
    assert not torch.equal(img, img_b)
    assert abs(float(img.mean()) - float(img_b.mean())) < 2e-3 * float(img.mean())


def test_full_size_parity_with_the_oracle_at_512x512x64(oracle):
    """BASELINE size against the oracle itself (not only properties): three randomised poses of the 53,248
    triangle vocal fold at 512x512, 64 spp — 50 M primary rays.  Primitive and shape ids must agree on EVERY
    ray (a box test that is not conservative shows up as a hit the oracle finds and the GPU misses), the
    image within 1e-4 of its scale with no pixel off by more than two flipped samples' worth."""
    sc = scenes.vocalfold()
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    go = oracle.Geometry(pool, tris, shape, off)
    cam = scene_desc.camera_from_sensor(sc.camera)
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    tex = _tex(sc)
    rng = np.random.default_rng(77)
    total = flips = lost = 0
    for i in range(3):
        a = rng.uniform(-0.2, 0.2)
        R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
        S = np.diag([rng.uniform(0.6, 1.6), rng.uniform(0.8, 1.2), 1.0, 1.0])
        T = np.eye(4)
        T[:3, 3] = rng.uniform(-0.08, 0.08, 3)
        xf = np.stack([np.eye(4), T @ R @ S]).astype(np.float32)
        offs = off.copy()
        offs[1] = off[1] + int(rng.integers(0, nfr[1])) * stride[1]
        gd.update(xf, offs)
        go.update(xf, offs)
        td, sd_, pd = gd.trace_primary(cam, 64, 1, seed=i)
        to, so, po = go.trace_primary(cam, 64, 1, seed=i)
        pd, sd_ = host(pd), host(sd_)
        total += pd.size
        flips += int(((pd != po) | (sd_ != so)).sum())
        lost += int(((pd < 0) & (po >= 0)).sum())
        same = pd == po
        np.testing.assert_allclose(host(td)[same], to[same], rtol=1e-5, atol=1e-6)
        img_d = host(gd.render_fwd(sd, dev(alb), tex, 64, seed=i))
        img_o = go.render_fwd(sd, alb, host(tex), 64, seed=i)
        _assert_image_close(img_d, img_o, 64, what=f"pose {i}")
        # ... and the kernel bench.py times: k_render_fwd_pk<1, wide, MATERIAL ROWS>, with the rows the bench's scene carries —
        # Mitsuba's principled BSDF under the reference's vocal-fold randomisation (examples/vocalfold_scene.py:86-93:
        # base colour between the two tissue tones, specular 0 .. 0.75; roughness 0.5) — and, on the last pose, near-mirror
        # rows (roughness 0.05: GGX alpha 0.0025, the case that used to differ by 0.4 % of a highlight)
        mats = np.zeros((2, 16), np.float32)
        mats[:, 0:3] = rng.uniform([0.8, 0.14, 0.34], [0.85, 0.5, 0.44], (2, 3))
        mats[:, 3] = 1.0
        mats[:, 4] = 0.5 if i < 2 else 0.05
        spec = rng.uniform(0.0, 0.75, 2)
        mats[:, 8] = 2.0 / (1.0 - np.sqrt(0.08 * spec)) - 1.0
        sdm = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16)
        img_d = host(gd.render_fwd(sdm, dev(mats), tex, 64, seed=i))
        img_o = go.render_fwd(sdm, mats, host(tex), 64, seed=i)
        _assert_image_close(img_d, img_o, 64, what=f"pose {i}, material rows")
        if i == 2:
            # ... and the launch the gradient bracket times: k_render_fwd_pk<1, wide, material rows, true> (ffx_render_fwd_adjoint, forward and
            # adjoint of a loss linear in the image in ONE launch) against the oracle's ffx_render_fwd_adjoint at the full size — image,
            # texture gradient and <gimg, img>.  gimg: the coverage loss's constant gradient plus a fixed pattern in red / blue, so that
            # every channel's albedo and colour factor is exercised.
            gimg = np.zeros((512, 512, 3), np.float32)
            gimg[..., 1] = -1.0 / (512 * 512)
            gimg[..., 0] = 0.5 / (512 * 512) * np.cos(np.arange(512, dtype=np.float32) * 0.05)[None, :]
            gimg[..., 2] = 0.25 / (512 * 512) * np.sin(np.arange(512, dtype=np.float32) * 0.03)[:, None]
            dot = torch.zeros(4096, device="cuda")
            img_f, gtex_f = gd.render_fwd_adjoint(sdm, dev(mats), tex, 64, i, dev(gimg), dot_out=dot)
            img_fo, gtex_o, dot_o = go.render_fwd_adjoint(sdm, mats, host(tex), 64, i, gimg)
            _assert_image_close(host(img_f), img_fo, 64, what="forward + adjoint launch at 512x512x64 (image)")
            np.testing.assert_allclose(host(img_f), img_d, rtol=0, atol=1e-6 * float(img_d.max()), err_msg="the forward + adjoint launch renders the plain forward's image")
            gs = float(np.abs(gtex_o).max())
            assert gs > 0
            ge = np.abs(host(gtex_f) - gtex_o)
            assert (ge > 1e-3 * gs).mean() <= 1e-3 and ge.max() <= 0.05 * gs, ((ge > 1e-3 * gs).mean(), ge.max() / gs)
            assert float(dot.double().sum()) == pytest.approx(dot_o, rel=1e-4)
            # ... and the gaussian film at the full size (round 4): ffx_render_fwd_filtered against the oracle's, the fused filtered forward +
            # adjoint launch against the filtered pair on the device (the oracle's filtered adjoint re-traces 16.8 M samples serially: minutes),
            # and the two directions as transposes of each other
            sdg = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16, rfilter="gaussian")
            img_g = gd.render_fwd(sdg, dev(mats), tex, 64, seed=i)
            _assert_image_close(host(img_g), go.render_fwd(sdg, mats, host(tex), 64, seed=i), 64, frac=5e-3, what="gaussian film at 512x512x64")
            gt_g = gd.render_bwd(sdg, dev(mats), 64, i, dev(gimg))
            img_gf, gt_gf = gd.render_fwd_adjoint(sdg, dev(mats), tex, 64, i, dev(gimg))
            assert torch.equal(img_gf, img_g)
            gsg = float(gt_g.abs().max())
            assert gsg > 0 and float((gt_gf - gt_g).abs().max()) <= 2e-4 * gsg
            base_g = gd.render_fwd(sdg, dev(mats), torch.zeros_like(tex), 64, seed=i)
            lhs = float(((img_g - base_g).double() * dev(gimg).double()).sum())
            rhs = float((tex.double() * gt_g[..., 0].double()).sum())
            assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)
            # ... the launches the NON-linear gradient bracket times (round-4 review, weak 2b): the cache-writing forward + K9
            # (k_render_fwd_pk with the footprint cache, k_render_bwd_cached_tiled16) under a sign-pattern gimg — what an L1 loss hands back —
            # against the oracle's own cache + cached adjoint at the full size
            sgn = np.where(np.random.default_rng(5).random((512, 512, 3)) < 0.5, -1.0, 1.0).astype(np.float32) / (512 * 512 * 3)
            cache = torch.zeros(ops.render_cache_bytes_sd(sdm, 64), dtype=torch.uint8, device="cuda")
            img_c = gd.render_fwd(sdm, dev(mats), tex, 64, seed=i, cache=cache)
            np.testing.assert_allclose(host(img_c), img_d, rtol=0, atol=1e-6 * float(img_d.max()))
            assert ops.render_cache_status(cache)[2] == 0
            gt_c = host(gd.render_bwd_cached(sdm, dev(mats), cache, 64, dev(sgn)))
            del cache
            _, cache_o = go.render_fwd_cache(sdm, mats, host(tex), 64, seed=i)
            gt_co = go.render_bwd_cached(sdm, mats, cache_o, 64, sgn)
            del cache_o
            gsc = float(np.abs(gt_co).max())
            gec = np.abs(gt_c - gt_co)
            assert gsc > 0 and (gec > 1e-3 * gsc).mean() <= 1e-3 and gec.max() <= 0.05 * gsc, ((gec > 1e-3 * gsc).mean(), gec.max() / gsc)
            # ... the filtered adjoints against the ORACLE at the full size (weak 2c; its re-trace is parallel since round 5), and the
            # reference-faithful non-linear step's pair (round 5): the filtered forward that stores per-sample records + the adjoint from them
            gt_go = go.render_bwd(sdg, mats, 64, i, sgn)
            gsg_o = float(np.abs(gt_go).max())
            gt_gs = host(gd.render_bwd(sdg, dev(mats), 64, i, dev(sgn)))
            ge_g = np.abs(gt_gs - gt_go)
            assert gsg_o > 0 and (ge_g > 1e-3 * gsg_o).mean() <= 2e-3 and ge_g.max() <= 0.05 * gsg_o, ((ge_g > 1e-3 * gsg_o).mean(), ge_g.max() / gsg_o)
            cache_g = torch.zeros(ops.render_cache_bytes_sd(sdg, 64), dtype=torch.uint8, device="cuda")
            img_gc = gd.render_fwd(sdg, dev(mats), tex, 64, seed=i, cache=cache_g)
            assert torch.equal(img_gc, img_g)
            gt_gc = host(gd.render_bwd_cached(sdg, dev(mats), cache_g, 64, dev(sgn), seed=i))
            del cache_g
            ge_gc = np.abs(gt_gc - gt_go)
            assert (ge_gc > 1e-3 * gsg_o).mean() <= 2e-3 and ge_gc.max() <= 0.05 * gsg_o, ((ge_gc > 1e-3 * gsg_o).mean(), ge_gc.max() / gsg_o)
            np.testing.assert_allclose(gt_gc, gt_gs, rtol=0, atol=2e-4 * gsg_o)  # (cached = re-traced up to the order of the float atomics)
    assert total == 3 * 512 * 512 * 64
    assert lost == 0, f"{lost} rays hit in the oracle and missed on the GPU"
    assert flips <= 2, f"{flips} of {total} rays hit a different primitive"


def test_k7_axis_parallel_rays_are_not_pathological():
    """Un-jittered pixel-corner rays of the centre row / column have a direction component of exactly
    0.  The box test must still reject boxes the origin lies outside of on that axis; when it did not
    (0 * inf = NaN dropped the axis) these rays walked the whole tree: 25 ms instead of 0.2 ms."""
    sc = scenes.vocalfold(frames=2)
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    cam = scene_desc.camera_from_sensor(sc.camera)

    def ms(jitter):
        gd.trace_primary(cam, 1, jitter, 1, want_ids=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            gd.trace_primary(cam, 1, jitter, 1, want_ids=False)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 3

    t_plain, t_jit = ms(0), ms(1)
    assert t_plain < 5.0 * t_jit + 0.5, (t_plain, t_jit)
    # and an exactly axis-parallel ray still reports the right hit
    o = dev(np.array([[0.0, 0.0, 1.5], [0.3, 0.0, 1.5]], np.float32))
    d = dev(np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], np.float32))
    t, s, p = gd.trace_rays(o, d)
    assert int(s[1]) >= 0 and float(t[1]) > 0  # the off-axis ray hits a fold


@pytest.mark.parametrize("env", [{"FFX_TRAVERSAL": "lane"}, {}, {"FFX_WIDE": "0"}, {"FFX_XCD_REMAP": "1", "FFX_PIXELS_PER_WAVE": "4"},
                                 {"FFX_WIDE": "0", "FFX_PIXELS_PER_WAVE": "1", "FFX_TILE_BLOCK": "0"}, {"FFX_XCD_REMAP": "16", "FFX_TILE_BLOCK": "2"},
                                 {"FFX_BINS": "0"}, {"FFX_BINS": "0", "FFX_WIDE": "0"}, {"FFX_BIN_CAP": "0"}, {"FFX_BIN_TILE": "4", "FFX_BIN_SPOT_N": "24"}])
def test_k8k9_every_kernel_variant_matches_the_oracle(oracle, env, monkeypatch):
    """the per-lane kernels (apex vectors formed per ray), the wave-packet kernels (apex records
    precomputed per render call) on the 64-wide walk (default) and on the binary walk (FFX_WIDE=0), and the
    launch-shape knobs all compute the same image and the same
    texture gradient (odd film size, spp not a multiple of 64, both shadow settings)."""
    # (FFX_BINS=0: the 64-wide / binary tree walks for every packet — the default until round 4 and still what a packet falls back to;
    # FFX_BIN_CAP=0: every grid's lists "overflow", so the pre-pass runs, marks the grids not-ok and the kernels take the fallback branch)
    for k in ("FFX_TRAVERSAL", "FFX_WIDE", "FFX_XCD_REMAP", "FFX_PIXELS_PER_WAVE", "FFX_TILE_BLOCK", "FFX_BINS", "FFX_BIN_CAP", "FFX_BIN_TILE", "FFX_BIN_SPOT_N"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.vocalfold(width=45, height=37, tex=64, frames=3, n_fold=20, tube=(20, 24))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 8))
    tex = _tex(sc, 1)
    rng = np.random.default_rng(2)
    gimg = rng.standard_normal((37, 45, 3)).astype(np.float32)
    for shadows, spp in ((True, 5), (False, 70)):
        sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=shadows)
        img_d = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=21))
        img_o = go.render_fwd(sd, alb, host(tex), spp, seed=21)
        scale, _ = _assert_image_close(img_d, img_o, spp, frac=1e-3, what=str(env))
        assert scale > 0.02
        gt_d = host(gd.render_bwd(sd, dev(alb), spp, 21, dev(gimg)))
        gt_o = go.render_bwd(sd, alb, spp, 21, gimg)
        gs = float(np.abs(gt_o).max())
        gerr = np.abs(gt_d - gt_o)
        assert gs > 0 and (gerr > 1e-3 * gs).mean() <= 1e-3 and gerr.max() <= 0.1 * gs, env


def _bin_headers(gd):
    """{ok, total, cap} of the three tile-bin grids (camera, projector, spot) of the blob the next render reads (ffx_common.h BinHdr)"""
    torch.cuda.synchronize()
    blob, info = gd.blob, gd.info
    out = []
    for a in range(3):
        o = int(info.off_bins) + a * int(info.bins_stride)
        out.append(tuple(int(v) for v in blob[o: o + 12].cpu().numpy().view(np.uint32)))
    return out


def test_tile_bins_that_overflow_or_cannot_be_built_fall_back_to_the_tree_walks(oracle, monkeypatch):
    """The paths behind the tile bins (round-4 review, weak 2a): (i) a grid whose entry lists do not fit its capacity (FFX_BIN_CAP, the
    test knob: in production a camera inside a coarse mesh) is marked not-ok by the pre-pass and ITS packets walk the tree while the other
    grids keep serving theirs — one render with some grids overflowed and some not; (ii) a spot light whose cone is too wide for a
    perspective grid (cutoff > 75 degrees) has no grid at all.  Either way: the oracle's image, gradient and hits, and bit for bit the
    image of the default path (the exact test alone decides hits)."""
    for k in ("FFX_BINS", "FFX_BIN_CAP", "FFX_TRAVERSAL", "FFX_WIDE"):
        monkeypatch.delenv(k, raising=False)
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.vocalfold(width=96, height=80, tex=96, frames=3, n_fold=24, tube=(24, 32))
    xf = _rand_xforms(2, 7)
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=xf)
    offs = gd._vert_off_host.copy()
    mats = material_rows(2, 17)
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16)
    cam = scene_desc.camera_from_sensor(sc.camera)
    tex = _tex(sc, 1)
    spp = 16
    rng = np.random.default_rng(4)
    gimg = rng.standard_normal((80, 96, 3)).astype(np.float32)
    img_ref = gd.render_fwd(sd, dev(mats), tex, spp, seed=5)
    hdrs = _bin_headers(gd)
    assert all(h[0] == 1 and 0 < h[1] <= h[2] for h in hdrs), hdrs  # all three grids built, lists within the capacity
    img_o = go.render_fwd(sd, mats, host(tex), spp, seed=5)
    g_o = go.render_bwd(sd, mats, spp, 5, gimg)
    gs = float(np.abs(g_o).max())
    t_o = go.trace_primary(cam, 4, 1, seed=9)
    totals = sorted(h[1] for h in hdrs)
    assert totals[0] < totals[2]
    for cap in (totals[0], totals[1], 0):  # the smallest list still fits / the two smaller ones / none
        monkeypatch.setenv("FFX_BIN_CAP", str(cap))
        gd.update(xf, offs)  # (a fresh pose: nothing of the previous capacity's pre-pass is claimed)
        img = gd.render_fwd(sd, dev(mats), tex, spp, seed=5)
        h2 = _bin_headers(gd)
        assert [h[0] for h in h2] == [int(h[1] <= cap) for h in hdrs] and [h[1] for h in h2] == [h[1] for h in hdrs], (cap, h2, hdrs)
        assert torch.equal(img, img_ref), f"cap {cap}: the image depends on which grids served it"
        _assert_image_close(host(img), img_o, spp, frac=2e-4, rel=1e-4, what=f"FFX_BIN_CAP={cap}")
        g_d = host(gd.render_bwd(sd, dev(mats), spp, 5, dev(gimg)))
        gerr = np.abs(g_d - g_o)
        assert gs > 0 and (gerr > 1e-3 * gs).mean() <= 1e-3 and gerr.max() <= 0.1 * gs, cap
        cache = torch.zeros(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
        assert torch.equal(gd.render_fwd(sd, dev(mats), tex, spp, seed=5, cache=cache), img_ref)
        g_c = host(gd.render_bwd_cached(sd, dev(mats), cache, spp, dev(gimg)))
        cerr = np.abs(g_c - g_o)
        assert (cerr > 1e-3 * gs).mean() <= 1e-3 and cerr.max() <= 0.1 * gs, cap
        td, sd_, pd = gd.trace_primary(cam, 4, 1, seed=9)
        _cmp_hits(td, sd_, pd, *t_o, f"K7, FFX_BIN_CAP={cap}")
    monkeypatch.delenv("FFX_BIN_CAP")
    # (ii) a spot wider than a perspective grid can hold: its shadow packets walk the tree, camera and projector keep their bins
    sdw = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16)
    sdw.spot.cutoff_deg, sdw.spot.beam_width_deg = 80.0, 60.0
    gd.update(xf, offs)
    img_w = host(gd.render_fwd(sdw, dev(mats), tex, spp, seed=5))
    hw = _bin_headers(gd)
    assert hw[0][0] == 1 and hw[1][0] == 1
    scale_w, _ = _assert_image_close(img_w, go.render_fwd(sdw, mats, host(tex), spp, seed=5), spp, frac=2e-4, rel=1e-4, what="spot cutoff 80 degrees")
    assert scale_w > 0.02 and np.abs(img_w - host(img_ref)).max() > 0.01 * scale_w  # (the wider cone lights more of the scene)
    monkeypatch.setenv("FFX_BINS", "0")
    gd.update(xf, offs)
    assert np.array_equal(host(gd.render_fwd(sdw, dev(mats), tex, spp, seed=5)), img_w)
    g_w = host(gd.render_bwd(sdw, dev(mats), spp, 5, dev(gimg)))
    g_wo = go.render_bwd(sdw, mats, spp, 5, gimg)
    gw = float(np.abs(g_wo).max())
    assert gw > 0 and (np.abs(g_w - g_wo) > 1e-3 * gw).mean() <= 1e-3


def _clear_fraction(gd, a):
    """share of the non-degenerate triangles whose flag word carries emitter a's "clear" bit (ffx_common.h FFX_GN_CLEAR_BIT: 1 projector, 2 spot)"""
    torch.cuda.synchronize()
    info, blob = gd.info, gd.blob
    F = int(info.n_tris)
    w = blob[int(info.off_gn): int(info.off_gn) + 16 * F].cpu().numpy().view(np.uint32).reshape(F, 4)[:, 3]
    ok = (w & 0x0FFFFFFF) != 0
    return float(((w[ok] >> (27 + a)) & 1).mean())


@pytest.mark.parametrize("which", ["vocalfold", "vocalfold_rows", "hello", "colon"])
def test_shadow_walks_skipped_for_clear_triangles_leave_the_image_as_it_is(oracle, which, monkeypatch):
    """Round 5 (an opt-in, FFX_SHADOW_CLEAR=1|2|3; off by default: it did not pay in the loop, ffx_trace.hip clear_enabled): the pre-pass
    proves, per triangle and emitter, that NOTHING can intersect a shadow segment ending on it (k_bin_clear: every
    triangle sharing a tile of the emitter's grid with it is apart in the emitter's image plane, behind its plane, or a front-facing
    neighbour whose plane it lies in front of) and a packet whose samples all lie on such triangles skips that emitter's any-hit stage —
    a quarter of the render kernel for the spot.  The proof is exact, so the image must not change by a bit: with and without the skip
    (FFX_SHADOW_CLEAR=0), on the smooth tube and folds (concave: H2), on a cube standing on a plane (convex edges, a concave crease, a
    cast shadow), on the colon; the texture gradient agrees to the order of the atomics; and the oracle's image is met."""
    for k in ("FFX_SHADOW_CLEAR", "FFX_BINS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("FFX_SHADOW_CLEAR", "3")  # both emitters (the default proves the spot only: the projector's stage is 3 % of the kernel)
    from tests.test_bruteforce_cpu import material_rows

    if which == "hello":
        sc, spp = scenes.hello_world(96, 80), 16
    elif which == "colon":
        sc, spp = scenes.colon(width=96, height=96, tex=128, n_around=48, n_along=160), 16
    else:
        sc, spp = scenes.vocalfold(width=96, height=80, tex=96, frames=3, n_fold=24, tube=(24, 32)), 16
    S = len(sc.meshes)
    xf = _rand_xforms(S, 12) if which != "hello" else None
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=xf)
    rows = which == "vocalfold_rows"
    mats = material_rows(S, 4) if rows else alb
    has_proj = sc.projector is not None
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16 if rows else 0)
    tex = _tex(sc, 1) if has_proj else None
    img_on = gd.render_fwd(sd, dev(mats), tex, spp, seed=9)
    frac_spot = _clear_fraction(gd, 2)
    assert 0.05 < frac_spot <= 1.0, frac_spot  # the proof succeeds somewhere ...
    if which == "hello":
        assert frac_spot < 1.0  # ... and fails where it must: the ground under the cube is shadowed
    g_on = gd.render_bwd(sd, dev(mats), spp, 9, torch.ones_like(img_on) / img_on.numel()) if has_proj else None
    monkeypatch.setenv("FFX_SHADOW_CLEAR", "0")
    pose = gd._vert_off_host.copy()
    gd.update(xf if xf is not None else np.tile(np.eye(4, dtype=np.float32), (S, 1, 1)), pose)
    img_off = gd.render_fwd(sd, dev(mats), tex, spp, seed=9)
    assert _clear_fraction(gd, 2) == 0.0 and _clear_fraction(gd, 1) == 0.0
    assert torch.equal(img_on, img_off), f"{which}: {int((img_on != img_off).sum())} pixel channels differ, worst {float((img_on - img_off).abs().max()):.3g}"
    if has_proj:
        g_off = gd.render_bwd(sd, dev(mats), spp, 9, torch.ones_like(img_on) / img_on.numel())
        assert float((g_on - g_off).abs().max()) <= 1e-4 * float(g_off.abs().max())
    img_o = go.render_fwd(sd, mats, host(tex) if has_proj else np.zeros((1, 1), np.float32), spp, seed=9)
    _assert_image_close(host(img_on), img_o, spp, frac=1e-3 if which == "hello" else 2e-4, what=which)


def _envelope(gd, a, n_cells_x, n_cells_y):
    """(header flag, [ny, nx, 4] cells) of emitter a's envelope in the blob the next render reads (ffx_common.h FFX_ENV_SUB, ffx_bin_off_env)"""
    torch.cuda.synchronize()
    info, blob = gd.info, gd.blob
    F, NT = int(info.n_tris), 16384
    off_entries = ((64 + 4 * (NT + 16) + 4 * NT + 63) // 64) * 64
    off_env = ((off_entries + 64 * (2 * F + NT + 64) + 63) // 64) * 64
    base = int(info.off_bins) + a * int(info.bins_stride)
    hdr = blob[base: base + 64].cpu().numpy().view(np.uint32)
    cells = blob[base + off_env: base + off_env + 16 * n_cells_x * n_cells_y].cpu().numpy().view(np.float32).reshape(n_cells_y, n_cells_x, 4)
    return int(hdr[4]), cells, int(hdr[0])


@pytest.mark.parametrize("which", ["vocalfold", "vocalfold_rows", "hello", "colon"])
def test_envelopes_settle_shadow_packets_and_leave_the_image_as_it_is(oracle, which, monkeypatch):
    """Round 6: the pre-pass leaves, per cell of an emitter's tile grid (7 x 7 cells per tile), ONE plane in front of every triangle the tile
    lists over that cell (k_bin_env), and a shadow packet all of whose segments end in front of their cells' planes skips the emitter's any-hit
    stage (bins_shadow) — 87 % of the packets of the vocal fold, K8 0.39 -> 0.34 ms.  The proof is conservative, so the image must not change
    by a bit: with the envelopes (default), without (FFX_ENVELOPE=0), and under the caller's hint that the renders are short
    (ffx_scene_desc.shadows | FFX_SHADOWS_PLAIN: the launch is left out, the header says so, the kernels do not look); on the smooth tube and
    folds, with material rows, on a cube standing on a plane (a cast shadow: cells that must NOT be proven), on the colon.  Forward, cache-writing
    forward, forward + adjoint in one launch; the re-traced gradient agrees to the order of the atomics; the oracle's image is met."""
    for k in ("FFX_ENVELOPE", "FFX_BINS", "FFX_SHADOW_CLEAR"):
        monkeypatch.delenv(k, raising=False)
    from tests.test_bruteforce_cpu import material_rows

    if which == "hello":
        sc, spp = scenes.hello_world(96, 80), 16
    elif which == "colon":
        sc, spp = scenes.colon(width=96, height=96, tex=128, n_around=48, n_along=160), 16
    else:
        sc, spp = scenes.vocalfold(width=96, height=80, tex=96, frames=3, n_fold=24, tube=(24, 32)), 16
    S = len(sc.meshes)
    xf = _rand_xforms(S, 12) if which != "hello" else None
    xfi = xf if xf is not None else np.tile(np.eye(4, dtype=np.float32), (S, 1, 1))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=xf)
    rows = which == "vocalfold_rows"
    mats = material_rows(S, 4) if rows else alb
    has_proj = sc.projector is not None
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16 if rows else 0)
    assert sd.shadows == 1
    tex = _tex(sc, 1) if has_proj else None
    pose = gd._vert_off_host.copy()
    gd.update(xfi, pose)
    img_on = gd.render_fwd(sd, dev(mats), tex, spp, seed=9)
    n_spot = min(128, max(8, int(2.0 * sd.spot.cutoff_deg + 0.999)))
    flag, cells, lists_ok = _envelope(gd, 2, 7 * n_spot, 7 * n_spot)
    # (the small colon's 15 k large triangles overflow the 120 x 120-tile grid of its 60-degree spot: its packets walk the tree, nobody builds or reads an envelope)
    assert flag == lists_ok and lists_ok == (0 if which == "colon" else 1)
    if lists_ok:
        finite = np.isfinite(cells[..., :3]).all(-1)
        planes = finite & (np.abs(cells[..., :3]).sum(-1) > 0)
        assert planes.mean() > 0.02 and (cells[..., 3] == 0).all(), planes.mean()  # cells that hold a plane (the rest: nothing listed, or no proof offered)
    if has_proj:
        assert _envelope(gd, 1, 7 * ((sd.proj.tex_w + 15) // 16), 7 * ((sd.proj.tex_h + 15) // 16))[0] == 1
    gimg = torch.ones_like(img_on) / img_on.numel()
    g_on = gd.render_bwd(sd, dev(mats), spp, 9, gimg) if has_proj else None
    fused_on = gd.render_fwd_adjoint(sd, dev(mats), tex, spp, 9, gimg) if has_proj else None
    # ---- without the envelopes: the same pose prepared again
    monkeypatch.setenv("FFX_ENVELOPE", "0")
    gd.update(xfi, pose)
    img_off = gd.render_fwd(sd, dev(mats), tex, spp, seed=9)
    assert _envelope(gd, 2, 8, 8)[0] == 0
    assert torch.equal(img_on, img_off), f"{which}: {int((img_on != img_off).sum())} pixel channels differ, worst {float((img_on - img_off).abs().max()):.3g}"
    if has_proj:
        g_off = gd.render_bwd(sd, dev(mats), spp, 9, gimg)
        assert float((g_on - g_off).abs().max()) <= 1e-4 * float(g_off.abs().max())
        fused_off = gd.render_fwd_adjoint(sd, dev(mats), tex, spp, 9, gimg)
        assert torch.equal(fused_on[0], fused_off[0]) and torch.equal(fused_on[0], img_on)
        assert float((fused_on[1] - fused_off[1]).abs().max()) <= 1e-4 * float(fused_off[1].abs().max())
    # ---- the caller's hint: short renders, no envelope launch — whatever the environment says
    monkeypatch.delenv("FFX_ENVELOPE")
    sdp = scene_desc.scene_desc(sc, tex_channels=1, shadows=3, mat_stride=16 if rows else 0)
    assert sdp.shadows == 3
    gd.update(xfi, pose)
    img_plain = gd.render_fwd(sdp, dev(mats), tex, spp, seed=9)
    assert _envelope(gd, 2, 8, 8)[0] == 0 and torch.equal(img_plain, img_on)
    # ... and a render that wants them on a pose prepared without: the call prepares its own (the key of what the blob holds differs)
    img_again = gd.render_fwd(sd, dev(mats), tex, spp, seed=9)
    assert _envelope(gd, 2, 8, 8)[0] == lists_ok and torch.equal(img_again, img_on)
    if has_proj and not rows:  # the cache-writing forward (keeps every projector walk: its stage is where the projector's envelope counts)
        cache = torch.zeros(ops.render_cache_bytes(sd.cam.width, sd.cam.height, spp), dtype=torch.uint8, device="cuda")
        assert torch.equal(gd.render_fwd(sd, dev(mats), tex, spp, seed=9, cache=cache), img_on)
    img_o = go.render_fwd(sd, mats, host(tex) if has_proj else np.zeros((1, 1), np.float32), spp, seed=9)
    _assert_image_close(host(img_on), img_o, spp, frac=1e-3 if which == "hello" else 2e-4, what=which)


@pytest.mark.parametrize("ch,k9_block", [(1, "16"), (1, "8"), (3, "16")])
def test_k9_cached_adjoint_matches_retrace_and_oracle(oracle, ch, k9_block, monkeypatch):
    """store-instead-of-retrace: the forward also writes, per pixel, the footprint of its samples in the
    projector texture (5x5 weights + window origin + shape; single samples that do not fit go to a small
    arena); the adjoint scatters the footprints.  Same image as the plain forward (bitwise), same gradient as
    the re-tracing adjoint and as the oracle (whose own cache keeps one record per sample — the cache is opaque,
    each library reads only what it wrote); it survives a re-fit between forward and backward; and it is an
    order of magnitude smaller than one record per sample."""
    monkeypatch.setenv("FFX_K9_BLOCK", k9_block)  # (1-channel textures: 16x16-pixel blocks by default, the round-2 8x8 kernel for A/B)
    sc = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    xf = _rand_xforms(2, 5)
    go, gd, alb = _pair(oracle, sc, frame=2, xforms=xf)
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True)
    tex = _tex(sc, ch)
    rng = np.random.default_rng(0)
    gimg = rng.standard_normal((44, 52, 3)).astype(np.float32)
    for spp in (9, 70):  # one pass; two passes of the same pixel (64 + 6 samples)
        nbytes = ops.render_cache_bytes(52, 44, spp)
        npx = 52 * 44  # header, 8-byte pixel headers, 112-byte footprints (each area padded to 128), 24-byte stray arena
        up = lambda v: ((v + 127) // 128) * 128  # noqa: E731
        assert nbytes == up(up(64 + 8 * npx) + 112 * npx) + 24 * max(4096, npx * spp // 64)
        cache = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        img_c = gd.render_fwd(sd, dev(alb), tex, spp, seed=3, cache=cache)
        img_p = gd.render_fwd(sd, dev(alb), tex, spp, seed=3)
        if os.environ.get("FFX_TRAVERSAL") == "lane":  # the cache is always written by the packet kernel
            torch.testing.assert_close(img_c, img_p, rtol=1e-4, atol=1e-5 * float(img_p.max()))
        else:
            assert torch.equal(img_c, img_p)
        hdr = host(cache[:12]).view(np.uint32)
        assert hdr[2] == 0 and hdr[0] <= hdr[1], f"stray arena: {hdr[0]} used of {hdr[1]}, {hdr[2]} dropped"
        img_o, cache_o = go.render_fwd_cache(sd, alb, host(tex), spp, seed=3)
        g_retrace = host(gd.render_bwd(sd, dev(alb), spp, 3, dev(gimg)))
        pose = (gd._vert_off_host.copy(),)
        gd.update(_rand_xforms(2, 99))  # re-fit to another pose: the cached adjoint must not care
        g_cached = host(gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg)))
        gd.update(xf, pose[0])
        g_oracle = go.render_bwd_cached(sd, alb, cache_o, spp, gimg)
        # the same launch can add <gimg, img> (the value of a linear loss) to per-block partial sums: both film formats, accumulating
        want = float((gimg.astype(np.float64) * host(img_c).astype(np.float64)).sum())
        mag = max(1.0, float(np.abs(gimg * host(img_c)).sum()))
        n_slots = ops.render_dot_slots(52, 44)
        assert n_slots == 7 * 6 and ops.render_dot_slots(512, 512) == 256  # one partial sum per 8x8-pixel block, folded onto at most 256
        dot = torch.full((n_slots,), 0.25, device="cuda")  # the slots are ADDED to
        g_dot = host(gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), img=img_c, dot_out=dot))
        assert abs(float(dot.double().sum()) - 0.25 * n_slots - want) <= 2e-5 * mag
        assert np.abs(g_dot - g_cached).max() <= 1e-3 * max(float(np.abs(g_cached).max()), 1e-20)  # (float atomics: order only)
        img16 = gd.render_fwd(sd, dev(alb), tex, spp, seed=3, fp16=True)
        dot16 = torch.zeros(n_slots, device="cuda")
        gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), img=img16, dot_out=dot16)
        want16 = float((gimg.astype(np.float64) * host(img16).astype(np.float64)).sum())
        assert abs(float(dot16.double().sum()) - want16) <= 2e-5 * mag
        _, dot_o = go.render_bwd_cached(sd, alb, cache_o, spp, gimg, img=img_o)
        assert abs(dot_o - want) <= 1e-3 * mag  # (the oracle's own image: equal within the image tolerance)
        with pytest.raises(ValueError):
            gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), img=img_c)
        with pytest.raises(ValueError):
            gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), img=img16, dot_out=torch.zeros(1, device="cuda"))
        # K9 under the reference's own loss, L1 against a target image (ffx_render_bwd_cached_l1, round 6): the scatter launch forms the loss's
        # gradient per pixel and emits the loss value into the slots — against ffx_l1_value_grad + ffx_render_bwd_cached and the oracle's composition
        # (the stray records replay the same per-pixel gradient)
        tgt = (host(img_c) + rng.standard_normal(host(img_c).shape).astype(np.float32) * 0.05).astype(np.float32)
        eq = rng.random(tgt.shape) < 0.1
        tgt[eq] = host(img_c)[eq]  # (exact ties: gradient 0)
        v_sep, g_sep = ops.l1_value_grad(img_c.reshape(-1), dev(tgt).reshape(-1), weight=0.7)
        gt_sep = host(gd.render_bwd_cached(sd, dev(alb), cache, spp, g_sep.view(img_c.shape)))
        slots = torch.zeros(n_slots, device="cuda")
        gt_l1 = torch.zeros((sd.proj.tex_h, sd.proj.tex_w, ch), device="cuda")
        res = gd.render_bwd_cached_l1(sd, dev(alb), cache, spp, img_c, dev(tgt), 0.7, gt_l1, slots)
        if ch == 1 and k9_block == "16":
            assert res is gt_l1
            assert np.abs(host(gt_l1) - gt_sep).max() <= 1e-4 * max(float(np.abs(gt_sep).max()), 1e-20)  # (float atomics: order only)
            assert float(slots.double().sum()) == pytest.approx(float(v_sep), rel=5e-6)
            gt_o, v_o = go.render_bwd_cached_l1(sd, alb, cache_o, spp, host(img_c), tgt, 0.7)
            err = np.abs(host(gt_l1) - gt_o)
            sc_o = float(np.abs(gt_o).max())
            assert (err > 1e-3 * sc_o).mean() <= 1e-3 and err.max() <= 0.1 * sc_o
            assert v_o == pytest.approx(float(v_sep), rel=1e-5)
        else:  # (three-channel textures and the 8x8-block kernel: declined, nothing written)
            assert res is None and float(gt_l1.abs().max()) == 0.0 and float(slots.abs().max()) == 0.0
        scale = float(np.abs(g_oracle).max())
        assert scale > 0
        for a, b in ((g_cached, g_retrace), (g_cached, g_oracle), (g_oracle, go.render_bwd(sd, alb, spp, 3, gimg))):
            err = np.abs(a - b)
            assert (err > 1e-3 * scale).mean() <= 1e-3 and err.max() <= 0.1 * scale
    # at the BASELINE size the cache is 37.7 MB (one 16-byte record per sample was 268 MB)
    assert ops.render_cache_bytes(512, 512, 64) <= 40 * 10**6


@pytest.mark.parametrize("ch", [1, 3])
@pytest.mark.parametrize("rows", ["albedo", "material_rows"])
def test_forward_and_adjoint_in_one_launch(oracle, ch, rows):
    """ffx_render_fwd_adjoint (round 3): for a loss whose gradient does not depend on the image the adjoint is formed inside the render —
    the pixel's footprint times gimg goes straight into gtex, stray samples at once, <gimg, img> into 4096 partial sums.  Same image as
    the plain forward (bitwise), same gradient as the cached and the re-tracing adjoints and as the oracle's composition; accumulates
    into the caller's gtex; one and two passes per pixel; the sparse flag agrees where the texture is not zero."""
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    go, gd, alb = _pair(oracle, sc, frame=2, xforms=_rand_xforms(2, 5))
    mats = alb if rows == "albedo" else material_rows(2, 31)
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=0 if rows == "albedo" else 16)
    tex = _tex(sc, ch)
    rng = np.random.default_rng(1)
    gimg = rng.standard_normal((44, 52, 3)).astype(np.float32)
    for spp in (9, 70):
        img_p = gd.render_fwd(sd, dev(mats), tex, spp, seed=3)
        dot = torch.full((4096,), 0.5, device="cuda")
        acc = torch.full((sd.proj.tex_h, sd.proj.tex_w, ch), 2.0, device="cuda")  # gtex is ACCUMULATED into
        img_f, g_f = gd.render_fwd_adjoint(sd, dev(mats), tex, spp, 3, dev(gimg), out=acc, dot_out=dot)
        assert torch.equal(img_f, img_p) and g_f is acc
        g_f = host(acc) - 2.0
        cache = torch.zeros(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
        gd.render_fwd(sd, dev(mats), tex, spp, seed=3, cache=cache)
        g_c = host(gd.render_bwd_cached(sd, dev(mats), cache, spp, dev(gimg)))
        g_r = host(gd.render_bwd(sd, dev(mats), spp, 3, dev(gimg)))
        img_o, g_o, dot_o = go.render_fwd_adjoint(sd, mats, host(tex), spp, 3, gimg)
        scale = float(np.abs(g_o).max())
        assert scale > 0
        for a, b, what in ((g_f, g_c, "cached"), (g_f, g_r, "re-traced"), (g_f, g_o, "oracle")):
            err = np.abs(a - b)
            assert (err > 1e-3 * scale).mean() <= 1e-3 and err.max() <= 0.1 * scale, (what, spp, float(err.max() / scale))
        want = float((gimg.astype(np.float64) * host(img_f).astype(np.float64)).sum())
        mag = max(1.0, float(np.abs(gimg * host(img_f)).sum()))
        assert abs(float(dot.double().sum()) - 0.5 * 4096 - want) <= 2e-5 * mag and abs(dot_o - want) <= 1e-3 * mag
        # fp16 film: <gimg, img> is taken with the image as stored
        dot16 = torch.zeros(4096, device="cuda")
        img16, _ = gd.render_fwd_adjoint(sd, dev(mats), tex, spp, 3, dev(gimg), dot_out=dot16, fp16=True)
        assert img16.dtype == torch.float16
        assert abs(float(dot16.double().sum()) - float((gimg.astype(np.float64) * host(img16).astype(np.float64)).sum())) <= 2e-5 * mag
    # sparse adjoint: texels whose value is zero may be left out, the others agree
    tex_s = tex.clone()
    tex_s[: tex_s.shape[0] // 2] = 0.0
    _, g_full = gd.render_fwd_adjoint(sd, dev(mats), tex_s, 9, 3, dev(gimg))
    _, g_sp = gd.render_fwd_adjoint(sd, dev(mats), tex_s, 9, 3, dev(gimg), sparse_adjoint=True)
    nz = (tex_s.reshape(g_full.shape) != 0)
    gs = float(g_full.abs().max())
    assert float(((g_full - g_sp).abs() * nz).max()) <= 2e-3 * gs
    with pytest.raises(ValueError):
        gd.render_fwd_adjoint(sd, dev(mats), tex, 9, 3, dev(gimg), dot_out=torch.zeros(7, device="cuda"))
    nop = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=0 if rows == "albedo" else 16)
    nop.proj.enabled = 0
    with pytest.raises(Exception, match="no projector"):
        gd.render_fwd_adjoint(nop, dev(mats), tex, 9, 3, dev(gimg))


def test_apex_records_written_ahead_and_cache_header_cleared_by_the_caller(oracle):
    """FFX_RENDER_APEX_READY / FFX_RENDER_CACHE_ZEROED (include/ffx.h): the apex records may be written by ffx_apex_prepare behind the
    re-fit (ops.DeviceGeometry.update(apex_sd=...)) and the cache header cleared by the caller, so that a render launches nothing in
    front of its kernel.  Same bits as the call that prepares for itself; the host-side bookkeeping (ops.apex_key) never claims
    records that a re-fit, a trace or a render from other positions has replaced."""
    sc = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    xf = _rand_xforms(2, 5)
    go, gd, alb = _pair(oracle, sc, frame=2, xforms=xf)
    offs = gd._vert_off_host.copy()
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    tex = _tex(sc, 1)
    i_cur = lambda: gd._cur if gd._async else 0  # noqa: E731
    assert gd._apex[i_cur()] is None  # a re-fit without apex_sd leaves nothing to claim
    ref = gd.render_fwd(sd, dev(alb), tex, 9, seed=3)  # (prepares for itself)
    assert gd._apex[i_cur()] == ops.apex_key(sd)
    assert torch.equal(gd.render_fwd(sd, dev(alb), tex, 9, seed=3), ref)  # second call on the same records: flag set, no pre-pass
    gd.update(xf, offs, apex_sd=sd)  # records re-written, apex records behind them on the side stream
    assert gd._apex[i_cur()] == ops.apex_key(sd)
    assert torch.equal(gd.render_fwd(sd, dev(alb), tex, 9, seed=3), ref)
    np.testing.assert_allclose(host(ref), go.render_fwd(sd, alb, host(tex), 9, seed=3), rtol=2e-4, atol=2e-6 * float(ref.max()))
    # another camera position between two renders of the first: each call sees its own records
    cam2 = np.array(sc.camera.to_world, np.float32).copy()
    cam2[:3, 3] += np.float32([0.01, -0.02, 0.015])
    sd2 = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, cam_to_world=cam2)
    other = gd.render_fwd(sd2, dev(alb), tex, 9, seed=3)
    assert gd._apex[i_cur()] == ops.apex_key(sd2) and not torch.equal(other, ref)
    np.testing.assert_allclose(host(other), go.render_fwd(sd2, alb, host(tex), 9, seed=3), rtol=2e-4, atol=2e-6 * float(ref.max()))
    assert torch.equal(gd.render_fwd(sd, dev(alb), tex, 9, seed=3), ref)
    # a primary trace from the camera of the last render finds the camera's records and tile bins in place (FFX_RENDER_APEX_READY in its
    # `jitter` argument): nothing is rewritten, the claim stays; from ANOTHER camera it rewrites the camera's area and claims that alone
    cam1 = scene_desc.camera_from_sensor(sc.camera)
    t_ready = gd.trace_primary(cam1, 2, 1, 5)
    assert gd._apex[i_cur()] == ops.apex_key(sd)
    cam2s = scene_desc.camera_from_sensor(sc.camera, cam2)
    t_other = gd.trace_primary(cam2s, 2, 1, 5)
    assert gd._apex[i_cur()] == ops.apex_key(cam=cam2s) != ops.apex_key(sd2)
    t_fresh = gd.trace_primary(cam1, 2, 1, 5)  # (writes the camera's area itself this time)
    assert gd._apex[i_cur()] == ops.apex_key(cam=cam1)
    for a_, b_ in zip(t_ready, t_fresh):
        assert torch.equal(a_, b_)
    assert not torch.equal(t_other[0], t_fresh[0])
    t_o = go.trace_primary(cam1, 2, 1, 5)
    same = (host(t_fresh[2]) == t_o[2])
    assert same.mean() > 0.998 and np.allclose(host(t_fresh[0])[same], t_o[0][same], rtol=2e-5, atol=2e-6)
    assert torch.equal(gd.render_fwd(sd, dev(alb), tex, 9, seed=3), ref)  # (a render re-derives its own: the emitters' part is not claimed by a trace)
    # the re-tracing adjoint prepares for itself and leaves its records behind
    gimg = dev(np.random.default_rng(0).standard_normal((44, 52, 3)).astype(np.float32))
    g_ref = gd.render_bwd(sd2, dev(alb), 9, 3, gimg)
    assert gd._apex[i_cur()] == ops.apex_key(sd2)
    assert torch.equal(gd.render_fwd(sd2, dev(alb), tex, 9, seed=3), other)
    # the cache header: cleared by the caller (FFX_RENDER_CACHE_ZEROED) or by the call — the same cache, the same adjoint
    nbytes = ops.render_cache_bytes(52, 44, 9)
    c_a = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")  # (a header full of garbage: the call resets it)
    c_b = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
    c_b[:64] = 0
    gd.update(xf, offs, apex_sd=sd2)
    img_a = gd.render_fwd(sd2, dev(alb), tex, 9, seed=3, cache=c_a)  # apex ready, header reset by the call (its own one-thread launch)
    img_b = gd.render_fwd(sd2, dev(alb), tex, 9, seed=3, cache=c_b, cache_zeroed=True)  # nothing in front of the kernel
    assert torch.equal(img_a, other) and torch.equal(img_b, other)
    ha, hb = host(c_a[:12]).view(np.uint32), host(c_b[:12]).view(np.uint32)
    assert (ha == hb).all() and ha[1] == max(4096, 52 * 44 * 9 // 64) and ha[2] == 0
    for c in (c_a, c_b):
        g = gd.render_bwd_cached(sd2, dev(alb), c, 9, gimg)
        assert float((g - g_ref).abs().max()) <= 2e-3 * float(g_ref.abs().max())
    # the raw entry point refuses what it cannot use
    with pytest.raises(Exception, match="apex_prepare"):
        gd._call("ffx_apex_prepare", None, None, None, None)


def test_adjoint_cache_overflow_is_refused_up_front_or_loud(oracle, monkeypatch):
    """The adjoint cache is lossy once its arena of single-sample records is full.  A projector texture much finer
    than the camera's pixels (55 texels per pixel here) makes most samples strays: (a) functional.cache_supported
    refuses the cache for such a render and the gradient comes from the re-tracing adjoint; (b) with the estimate
    overridden the forward drops samples — ffx_render_cache_status reports them, ffx_render_bwd_cached poisons gtex[0]
    with NaN instead of returning a gradient with holes, and the autograd path re-traces (or raises once the scene has
    been re-fitted)."""
    from fireflies_amd import functional as Fn

    sc = scenes.vocalfold(width=40, height=32, tex=1024, frames=3, n_fold=20, tube=(20, 24))
    xf = _rand_xforms(2, 5)
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=xf)
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    spp = 64
    tex = torch.rand(1024, 1024, device="cuda") + 0.1
    gimg = dev(np.random.default_rng(0).standard_normal((32, 40, 3)).astype(np.float32))
    monkeypatch.delenv("FFX_CACHE_MAX_TEXELS_PER_PIXEL", raising=False)
    assert Fn.texels_per_pixel(sd) > 50 and not Fn.cache_supported(sd, spp)
    g_oracle = go.render_bwd(sd, alb, spp, 3, host(gimg))[..., 0]
    scale = float(np.abs(g_oracle).max())

    def grad_through_autograd():
        leaf = tex.clone().requires_grad_(True)
        img = Fn.render(leaf, gd, sd, dev(alb), spp, seed=3)
        (g,) = torch.autograd.grad((img * gimg).sum(), leaf)
        return host(g)

    def close(g):
        err = np.abs(g - g_oracle)
        return np.isfinite(g).all() and (err > 1e-3 * scale).mean() <= 1e-3 and err.max() <= 0.1 * scale

    assert scale > 0 and close(grad_through_autograd())  # (a): re-traced
    # (b) force the cache
    monkeypatch.setenv("FFX_CACHE_MAX_TEXELS_PER_PIXEL", "1e9")
    assert Fn.cache_supported(sd, spp)
    cache = torch.zeros(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
    img_c = gd.render_fwd(sd, dev(alb), tex.unsqueeze(-1), spp, seed=3, cache=cache)
    assert torch.equal(img_c, gd.render_fwd(sd, dev(alb), tex.unsqueeze(-1), spp, seed=3))  # the image does not depend on the cache
    used, cap, dropped = ops.render_cache_status(cache)
    assert cap == 4096 and dropped > 0 and used >= cap, (used, cap, dropped)
    g_holes = gd.render_bwd_cached(sd, dev(alb), cache, spp, gimg)
    assert bool(torch.isnan(g_holes.reshape(-1)[0]))  # poisoned, not silently wrong
    assert close(grad_through_autograd())  # the autograd path saw `dropped` and re-traced
    leaf = tex.clone().requires_grad_(True)
    img = Fn.render(leaf, gd, sd, dev(alb), spp, seed=3)
    gd.update(_rand_xforms(2, 99))  # re-fitted between forward and backward: nothing to re-trace
    with pytest.raises(Fn.CacheOverflowError):
        torch.autograd.grad((img * gimg).sum(), leaf)
    # a cache that fits reports dropped == 0
    sc2 = scenes.vocalfold(width=40, height=32, tex=64, frames=3, n_fold=20, tube=(20, 24))
    sd2 = scene_desc.scene_desc(sc2, tex_channels=1, shadows=True)
    cache2 = torch.zeros(ops.render_cache_bytes_sd(sd2, 8), dtype=torch.uint8, device="cuda")
    gd.update(xf)
    gd.render_fwd(sd2, dev(alb), torch.rand(64, 64, 1, device="cuda"), 8, seed=3, cache=cache2)
    assert ops.render_cache_status(cache2)[2] == 0
    # FFX_RENDER_CACHE_KEEP_DROPPED (ABI 7; round-4 advisor): a step that reuses ONE cache for its scene samples must still know at its end
    # that an EARLIER sample overflowed — the later samples' resets empty the arena but keep the count; a plain call clears it
    sd3 = scene_desc.scene_desc(scenes.vocalfold(width=40, height=32, tex=64, frames=3, n_fold=20, tube=(20, 24)), tex_channels=1, shadows=True)
    assert ops.render_cache_bytes_sd(sd3, spp) == ops.render_cache_bytes_sd(sd, spp)  # (same film, same spp: the same cache layout)
    gd.render_fwd(sd, dev(alb), tex.unsqueeze(-1), spp, seed=3, cache=cache)  # sample 0: overflows
    d0 = ops.render_cache_status(cache)[2]
    assert d0 > 0
    tex3 = torch.rand(64, 64, 1, device="cuda")
    gd.render_fwd(sd3, dev(alb), tex3, spp, seed=4, cache=cache, keep_dropped=True)  # sample 1: fits, but the step's count stays
    used1, cap1, d1 = ops.render_cache_status(cache)
    assert d1 == d0 and used1 < cap1
    assert bool(torch.isnan(gd.render_bwd_cached(sd3, dev(alb), cache, spp, gimg).reshape(-1)[0]))  # ... and K9 keeps poisoning the step's gradient
    gd.render_fwd(sd3, dev(alb), tex3, spp, seed=4, cache=cache)  # the next step's first sample: a clean header
    assert ops.render_cache_status(cache)[2] == 0
    assert bool(torch.isfinite(gd.render_bwd_cached(sd3, dev(alb), cache, spp, gimg)).all())


# ------------------------------------------------------------------ fused pattern side of an optimisation step
@pytest.mark.parametrize("n,size,sigma", [(64, (96, 80), 10.0), (256, (500, 500), 10.0), (700, (128, 128), 30.0)])
def test_fused_pattern_kernels_match_the_unfused_oracle(oracle, n, size, sigma):
    """ffx_pattern_fwd / ffx_pattern_bwd / ffx_adam_clamp_step (three launches) against the oracle's composition of
    the separate entry points, which follow the reference line by line (laser.py:262-275, rasterization.py:7-37,
    156-161,589-600) — and the Adam part against torch.optim.Adam itself."""
    rng = np.random.default_rng(n)
    s0, s1 = size
    g2 = load_golden("g2_projection.npz")
    KF = (g2["K"] @ FLIP_Y).astype(np.float32)
    ndc = (rng.random((n, 3)) * np.array([0.9, 0.9, 0.0]) + np.array([0.05, 0.05, -1.0])).astype(np.float32)
    rays = oracle.transform_points(ndc, np.linalg.inv(KF.astype(np.float64)).astype(np.float32))
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    pts_o, tsum_o, tsor_o, ws_o = oracle.pattern_fwd(rays, KF, sigma, s0, s1, True)
    pts_d, tsum_d, tsor_d, ws_d = ops.pattern_fwd(dev(rays), KF, sigma, s0, s1, True)
    np.testing.assert_allclose(host(pts_d), pts_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(tsum_d), tsum_o, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(tsor_d), tsor_o, rtol=2e-6, atol=1e-6)
    assert float(host(ws_d).sum()) == pytest.approx(float(ws_o.sum()), rel=1e-5)
    # the fused forward equals the separate HIP kernels bit for bit (same tiles, same point order)
    assert torch.equal(tsum_d, ops.splat_fwd(pts_d, sigma, "sum", -1, s0, s1)) and torch.equal(tsor_d, ops.splat_fwd(pts_d, sigma, "softor", -1, s0, s1))
    # the same launch clears a caller's buffer (the optimisation step's gradient + loss accumulator): any length, nothing else touched
    for nz in (1, s0 * s1 + 1, 3 * s0 * s1 + 7):
        buf = torch.full((nz + 2,), 7.0, device="cuda")
        _, tsum_z, _, _ = ops.pattern_fwd(dev(rays), KF, sigma, s0, s1, True, zero=buf[1:-1])
        assert torch.equal(tsum_z, tsum_d) and float(buf[0]) == 7.0 and float(buf[-1]) == 7.0 and float(buf[1:-1].abs().max()) == 0.0
    gts = rng.standard_normal((s1, s0)).astype(np.float32)
    for w in (0.1, 0.0):
        gd_o, gr_o, val_o = oracle.pattern_bwd(rays, KF, sigma, s0, s1, tsum_o, tsor_o, gts, w, ws_o)
        gd_d, gr_d, val_d = ops.pattern_bwd(dev(rays), KF, sigma, s0, s1, tsum_d, tsor_d, dev(gts), w, ws_d)
        np.testing.assert_allclose(host(gd_d), gd_o, rtol=2e-4, atol=2e-5 * np.abs(gd_o).max())
        if w > 0:
            np.testing.assert_allclose(host(gr_d), gr_o, rtol=5e-4, atol=5e-5 * np.abs(gr_o).max())
            assert float(val_d[0]) == pytest.approx(val_o, rel=1e-5)
        else:
            assert gr_d is None and float(val_d[0]) == 0.0
    # the data term's partial sums (K9's dot slots) are summed by the same launch: total loss and the raw sum
    li = rng.standard_normal(300).astype(np.float32)
    _, _, val_l = ops.pattern_bwd(dev(rays), KF, sigma, s0, s1, tsum_d, tsor_d, dev(gts), 0.1, ws_d, loss_in=dev(li), loss_div=4.0)
    _, _, reg_o, (tot_o, raw_o) = oracle.pattern_bwd(rays, KF, sigma, s0, s1, tsum_o, tsor_o, gts, 0.1, ws_o, loss_in=li, loss_div=4.0)
    assert float(val_l[2]) == pytest.approx(float(li.astype(np.float64).sum()), abs=1e-4) and raw_o == pytest.approx(float(val_l[2]), abs=1e-4)
    assert float(val_l[1]) == pytest.approx(tot_o, rel=1e-5, abs=1e-5) and float(val_l[0]) == pytest.approx(reg_o, rel=1e-5)
    # no data term (a rank without samples): only the regulariser
    gd_d, gr_d, _ = ops.pattern_bwd(dev(rays), KF, sigma, s0, s1, tsum_d, tsor_d, None, 0.1, ws_d)
    assert gd_d is None
    np.testing.assert_allclose(host(gr_d), gr_o if False else oracle.pattern_bwd(rays, KF, sigma, s0, s1, tsum_o, tsor_o, None, 0.1, ws_o)[1], rtol=5e-4,
                               atol=5e-5 * np.abs(gr_o).max() if np.abs(gr_o).max() > 0 else 1e-9)
    # Adam + clamp_to_fov + normalise: three steps against the oracle and against torch.optim.Adam
    KFi = np.linalg.inv(KF.astype(np.float64)).astype(np.float32)
    r_d = dev(rays).clone()
    m_d, v_d, st_d = torch.zeros_like(r_d), torch.zeros_like(r_d), torch.zeros((), device="cuda")
    r_o, m_o, v_o, st_o = rays.copy(), np.zeros_like(rays), np.zeros_like(rays), np.zeros(1, np.float32)
    r_t = dev(rays).clone().requires_grad_(True)
    opt = torch.optim.Adam([r_t], lr=5e-3)
    for k in range(3):
        g = (rng.standard_normal(rays.shape) * 0.3).astype(np.float32)
        ops.adam_clamp_step_(r_d, dev(g), m_d, v_d, st_d, 5e-3, 0.9, 0.999, 1e-8, KF, KFi, 0.05, 0.95, 2)
        oracle.adam_clamp_step(r_o, g, m_o, v_o, st_o, 5e-3, 0.9, 0.999, 1e-8, KF, KFi, 0.05, 0.95, 2)
        r_t.grad = dev(g)
        opt.step()
        with torch.no_grad():
            ops.clamp_to_fov_(r_t.detach(), KF, KFi, 0.05, 0.95, 2)
        np.testing.assert_allclose(host(r_d), r_o, rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(host(r_d), host(r_t), rtol=2e-6, atol=2e-7)
    assert float(st_d) == 3.0 and st_o[0] == 3.0
    np.testing.assert_allclose(host(m_d), m_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(host(m_d), host(opt.state[r_t]["exp_avg"]), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(host(v_d), host(opt.state[r_t]["exp_avg_sq"]), rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("n,size,sigma,ks", [(64, (96, 80), 10.0, 5), (256, (500, 500), 10.0, 5), (700, (128, 128), 30.0, 5), (40, (70, 33), 10.0, 3),
                                             (20, (64, 64), 400.0, 5), (9, (6, 40), 4.0, 5)])
def test_pattern_launches_that_carry_the_blur_and_the_update_along(oracle, n, size, sigma, ks):
    """ffx_pattern_fwd_blur / ffx_pattern_bwd_blur (round 3): the texture finalise K3 rides on the splat's tiles, its transpose is applied
    inside the gradient launch over the points' footprints only, and the Adam + clamp_to_fov update is done by the workgroup that
    finishes last.  Bit for bit the separate launches' results (same arithmetic in the same order: that is the design), and the oracle's
    composition within the tolerances of the separate kernels.  Cases: border points, a 3x3 kernel (run-time size), a footprint too
    large for the workgroup's LDS and an image narrower than the kernel's reflections (both fall back to the transpose blur as its own
    launch), points whose footprints hang over the image border."""
    rng = np.random.default_rng(n)
    s0, s1 = size
    g2 = load_golden("g2_projection.npz")
    KF = (g2["K"] @ FLIP_Y).astype(np.float32)
    KFi = np.linalg.inv(KF.astype(np.float64)).astype(np.float32)
    ndc = (rng.random((n, 3)) * np.array([1.0, 1.0, 0.0]) + np.array([0.0, 0.0, -1.0])).astype(np.float32)  # up to the very border
    rays = oracle.transform_points(ndc, KFi)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    bs = 3.0
    # ---- forward
    pts_a, tsum_a, tsor_a, ws_a = ops.pattern_fwd(dev(rays), KF, sigma, s0, s1, True)
    tex_a = ops.blur_fwd(tsum_a, ks, bs)
    buf = torch.full((s0 * s1 + 9,), 7.0, device="cuda")
    pts_b, tsum_b, tsor_b, ws_b, tex_b = ops.pattern_fwd_blur(dev(rays), KF, sigma, s0, s1, ks, bs, True, zero=buf[1:-1])
    for a, b, what in ((pts_a, pts_b, "pts"), (tsum_a, tsum_b, "tsum"), (tsor_a, tsor_b, "tsor"), (ws_a, ws_b, "ws"), (tex_a, tex_b, "tex")):
        assert torch.equal(a, b), what
    assert float(buf[0]) == 7.0 and float(buf[-1]) == 7.0 and float(buf[1:-1].abs().max()) == 0.0
    _, tsum_n, tsor_n, ws_n, tex_n = ops.pattern_fwd_blur(dev(rays), KF, sigma, s0, s1, ks, bs, False)  # without the regulariser's outputs
    assert tsor_n is None and ws_n is None and torch.equal(tsum_n, tsum_a) and torch.equal(tex_n, tex_a)
    pts_o, tsum_o, tsor_o, ws_o, tex_o = oracle.pattern_fwd_blur(rays, KF, sigma, s0, s1, ks, bs, True)
    np.testing.assert_allclose(host(tex_b), tex_o, rtol=3e-6, atol=2e-6)
    # ---- backward: K3^T inside the gradient launch
    gtex = rng.standard_normal((s1, s0)).astype(np.float32)
    gtex[rng.random((s1, s0)) < 0.3] = 0.0
    li = rng.standard_normal(37).astype(np.float32)
    for w in (0.1, 0.0):
        gts = ops.blur_bwd(dev(gtex), ks, bs)
        gd_a, gr_a, val_a = ops.pattern_bwd(dev(rays), KF, sigma, s0, s1, tsum_a, tsor_a, gts, w, ws_a, loss_in=dev(li), loss_div=2.0)
        scratch = torch.empty_like(tsum_a)
        gd_b, gr_b, val_b = ops.pattern_bwd_blur(dev(rays), KF, sigma, s0, s1, tsum_a, tsor_a, dev(gtex), w, ws_a, ks, bs, loss_in=dev(li), loss_div=2.0, scratch=scratch)
        assert torch.equal(gd_a, gd_b) and torch.equal(val_a, val_b) and (gr_a is None) == (gr_b is None) and (gr_a is None or torch.equal(gr_a, gr_b))
        gd_o, gr_o, val_o = oracle.pattern_bwd_blur(rays.copy(), KF, sigma, s0, s1, tsum_o, tsor_o, gtex, w, ws_o, ks, bs, loss_in=li, loss_div=2.0)
        np.testing.assert_allclose(host(gd_b), gd_o, rtol=3e-4, atol=3e-5 * max(np.abs(gd_o).max(), 1e-20))
        if w > 0:
            np.testing.assert_allclose(host(gr_b), gr_o, rtol=5e-4, atol=5e-5 * max(np.abs(gr_o).max(), 1e-20))
        np.testing.assert_allclose(host(val_b), val_o, rtol=2e-5, atol=2e-5)
    # ksize 0: gtex is the gradient on the sum texture itself (ffx_pattern_bwd's meaning)
    gd_c, _, _ = ops.pattern_bwd_blur(dev(rays), KF, sigma, s0, s1, tsum_a, tsor_a, dev(gtex), 0.0, ws_a, 0, 1.0)
    assert torch.equal(gd_c, ops.pattern_bwd(dev(rays), KF, sigma, s0, s1, tsum_a, tsor_a, dev(gtex), 0.0, ws_a)[0])
    # ---- the update carried along: three steps, against the separate launches (bitwise) and the oracle
    def fresh():
        r = dev(rays).clone()
        return r, torch.zeros_like(r), torch.zeros_like(r), torch.zeros((), device="cuda")

    (r_a, m_a, v_a, st_a), (r_b, m_b, v_b, st_b) = fresh(), fresh()
    r_o, m_o, v_o, st_o = rays.copy(), np.zeros_like(rays), np.zeros_like(rays), np.zeros(1, np.float32)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k in range(3):
        gtex_k = (rng.standard_normal((s1, s0)) * (k + 1)).astype(np.float32)
        w = 0.1 if k != 1 else 0.0  # (a step without the regulariser in between: grays_reg is then not combined)
        _, ts_k, to_k, ws_k, _ = ops.pattern_fwd_blur(r_a, KF, sigma, s0, s1, ks, bs, True)
        gd, gr, _ = ops.pattern_bwd(r_a, KF, sigma, s0, s1, ts_k, to_k, ops.blur_bwd(dev(gtex_k), ks, bs), w, ws_k)
        g_a = torch.empty_like(r_a)
        ops.adam_clamp_step_(r_a, gd, m_a, v_a, st_a, 5e-3, 0.9, 0.999, 1e-8, KF, KFi, 0.05, 0.95, 2, grad_b=gr, grad_div=4.0, grad_out=g_a)
        _, ts_b, to_b, ws_kb, _ = ops.pattern_fwd_blur(r_b, KF, sigma, s0, s1, ks, bs, True)
        g_b = torch.empty_like(r_b)
        aa = ops.adam_args(r_b, m_b, v_b, st_b, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=g_b)
        ops.pattern_bwd_blur(r_b, KF, sigma, s0, s1, ts_b, to_b, dev(gtex_k), w, ws_kb, ks, bs, adam=aa, scratch=torch.empty_like(ts_b))
        assert torch.equal(r_a, r_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b) and torch.equal(g_a, g_b) and float(st_b) == k + 1 and int(counter) == 0
        _, ts_ok, to_ok, ws_ok, _ = oracle.pattern_fwd_blur(r_o, KF, sigma, s0, s1, ks, bs, True)
        _, _, _, g_o = oracle.pattern_bwd_blur(r_o, KF, sigma, s0, s1, ts_ok, to_ok, gtex_k, w, ws_ok, ks, bs,
                                               adam=dict(exp_avg=m_o, exp_avg_sq=v_o, step=st_o, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8, KF_inv=KFi, lo=0.05, hi=0.95,
                                                         grad_div=4.0, n_normalize=2))
        np.testing.assert_allclose(host(g_b), g_o, rtol=1e-3, atol=1e-4 * max(np.abs(g_o).max(), 1e-20))
    np.testing.assert_allclose(host(r_b), r_o, rtol=1e-4, atol=2e-5)  # (Adam's first steps are sign-like: tiny gradient differences move a ray by lr at most)
    # ---- the data term as an inner product evaluated by the same launch (ffx_adam_args.dot_*): <a, b> over an odd number of floats,
    #      16-byte aligned and not; reg_value[2] = the sum, reg_value[1] = sum / loss_div + regulariser — and the update is the same
    #      b may be shorter than a and is then repeated (the renders of a step stacked against one constant gradient)
    for n_dot, shift, reps in ((3 * 37 * 29, 0, 1), (4099, 1, 1), (5, 0, 1), (4 * 500, 0, 3), (7 * 11, 0, 5)):
        b_np = rng.standard_normal(n_dot + shift).astype(np.float32)
        a_np = np.concatenate([np.zeros(shift, np.float32), rng.standard_normal(n_dot * reps).astype(np.float32)])
        a_d, b_d = dev(a_np)[shift:], dev(b_np)[shift:]
        b_np = np.concatenate([b_np[:shift], np.tile(b_np[shift:], reps)])  # (what the products are taken with)
        (r_c, m_c, v_c, st_c), (r_e, m_e, v_e, st_e) = fresh(), fresh()
        _, ts_c, to_c, ws_c, _ = ops.pattern_fwd_blur(r_c, KF, sigma, s0, s1, ks, bs, True)
        part = torch.empty(n, device="cuda")
        # (grad_out as a temporary: ops.adam_args keeps the tensors it takes addresses of alive — a freed one used to be handed to the NEXT small
        # allocation, the launch's 3-float value buffer, which the update's gradient store then overwrote: val_c[0] came back as a gradient)
        aa = ops.adam_args(r_c, m_c, v_c, st_c, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=torch.empty_like(r_c), dot=(a_d, b_d, part))
        _, _, val_c = ops.pattern_bwd_blur(r_c, KF, sigma, s0, s1, ts_c, to_c, dev(gtex), 0.1, ws_c, ks, bs, loss_div=4.0, adam=aa, scratch=torch.empty_like(ts_c))
        aa = ops.adam_args(r_e, m_e, v_e, st_e, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=torch.empty_like(r_e))
        _, _, val_e = ops.pattern_bwd_blur(r_e, KF, sigma, s0, s1, ts_c, to_c, dev(gtex), 0.1, ws_c, ks, bs, loss_div=4.0, adam=aa, scratch=torch.empty_like(ts_c))
        want = float((a_np[shift:].astype(np.float64) * b_np[shift:].astype(np.float64)).sum())
        mag = float(np.abs(a_np[shift:] * b_np[shift:]).sum())
        assert abs(float(val_c[2]) - want) <= 2e-6 * mag and float(val_c[0]) == float(val_e[0])
        assert float(val_c[1]) == pytest.approx(want / 4.0 + float(val_c[0]), rel=1e-5, abs=1e-6 * mag)
        assert torch.equal(r_c, r_e) and int(counter) == 0
        # Adam arguments WITHOUT state: no update, only the inner product (a multi-rank step exchanges the gradient before it updates)
        r_n = dev(rays).clone()
        aa = ops.adam_args(r_n, None, None, None, counter, 0.0, 0.0, 0.0, 0.0, KFi, 0.0, 1.0, dot=(a_d, b_d, part))
        gd_n, _, val_n = ops.pattern_bwd_blur(r_n, KF, sigma, s0, s1, ts_c, to_c, dev(gtex), 0.1, ws_c, ks, bs, loss_div=4.0, adam=aa, scratch=torch.empty_like(ts_c))
        assert torch.equal(r_n, dev(rays)) and float(val_n[2]) == float(val_c[2]) and int(counter) == 0
        with pytest.raises(Exception, match="not both"):
            bad_aa = ops.adam_args(r_c, m_c, v_c, st_c, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=torch.empty_like(r_c), dot=(a_d, b_d, part))
            ops.pattern_bwd_blur(r_c, KF, sigma, s0, s1, ts_c, to_c, dev(gtex), 0.1, ws_c, ks, bs, loss_in=dev(li), adam=bad_aa)
    _, _, val_o, _ = oracle.pattern_bwd_blur(rays.copy(), KF, sigma, s0, s1, tsum_o, tsor_o, gtex, 0.1, ws_o, ks, bs, loss_div=4.0,
                                              adam=dict(exp_avg=np.zeros_like(rays), exp_avg_sq=np.zeros_like(rays), step=np.zeros(1, np.float32), lr=5e-3, beta1=0.9, beta2=0.999,
                                                        eps=1e-8, KF_inv=KFi, lo=0.05, hi=0.95, grad_div=4.0, n_normalize=2, dot=(a_np[shift:], b_np[shift:shift + n_dot])))
    assert abs(float(val_o[2]) - want) <= 2e-6 * max(mag, 1.0)
    # misuse is refused: an update of other rays than the gradient's, Adam arguments without a counter
    bad = ops.adam_args(r_a, m_b, v_b, st_b, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=g_b)
    with pytest.raises(Exception, match="rays the gradient was taken at"):
        ops.pattern_bwd_blur(r_b, KF, sigma, s0, s1, ts_b, to_b, dev(gtex), 0.1, ws_kb, ks, bs, adam=bad)
    with pytest.raises(Exception, match="ksize"):
        ops.pattern_fwd_blur(dev(rays), KF, sigma, s0, s1, 4, bs, True)


@pytest.mark.parametrize("n,size,sigma", [(64, (500, 500), 10.0), (64, (96, 80), 10.0), (300, (250, 120), 6.0), (9, (40, 24), 4.0), (1500, (1024, 1024), 10.0)])
def test_the_pattern_side_of_a_step_as_one_launch(oracle, n, size, sigma):
    """ffx_pattern_step (round 6): ffx_pattern_bwd_blur(..., adam) of one step and ffx_pattern_fwd_blur of the next in ONE launch — workgroups take
    their roles by ticket, helpers wait for the update, then splat and blur the updated pattern and clear the accumulator the gradient was read from.
    Five consecutive steps (so that the sync words are re-armed by the launch itself and the running products beta^t take over from pow()) against
    the two separate launches, BIT FOR BIT: gradient, loss values, rays, both moments, step count, the combined gradient, and the next step's
    pts / tsum / tsor / ws / tex; the cleared range exactly the range asked for; the data term as partial sums and as an inner product; a guard that
    skips the update (the copy of its header in the sync words, the header itself cleared); the kept pattern and the `stale` word; and the first
    step's gradient against the oracle.  Sizes: the bench's, a small image, more points than a workgroup has threads, a tiny image, and more points +
    tiles than the GPU holds workgroups at a time (1500 + 1024 workgroups of which helpers take four tiles each)."""
    rng = np.random.default_rng(n + size[0])
    s0, s1 = size
    g2 = load_golden("g2_projection.npz")
    KF = (g2["K"] @ FLIP_Y).astype(np.float32)
    KFi = np.linalg.inv(KF.astype(np.float64)).astype(np.float32)
    ndc = (rng.random((n, 3)) * np.array([1.0, 1.0, 0.0]) + np.array([0.0, 0.0, -1.0])).astype(np.float32)
    rays = oracle.transform_points(ndc, KFi)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    ks, bs = 5, 3.0
    T = s0 * s1

    def fresh():
        r = dev(rays).clone()
        return r, torch.zeros_like(r), torch.zeros_like(r), torch.zeros((), device="cuda")

    for variant in ("slots", "dot", "guard"):
        (r_a, m_a, v_a, st_a), (r_b, m_b, v_b, st_b) = fresh(), fresh()
        counter = torch.zeros(1, dtype=torch.int32, device="cuda")
        sync = torch.zeros(35840, dtype=torch.uint8, device="cuda")
        kept = torch.zeros((2,) + tuple(r_b.shape), device="cuda")
        # the accumulator as optim.PatternOptimizer lays it out: gtex, loss slots, a 64-byte header behind them — plus a canary on either side
        n_slots = 37
        acc_a = torch.zeros(1 + T + n_slots + 16 + 1, device="cuda")
        acc_b = torch.zeros_like(acc_a)
        n_dot = 3 * 41 * 29
        a_d, b_d = dev(rng.standard_normal(n_dot).astype(np.float32)), dev(rng.standard_normal(n_dot).astype(np.float32))
        part_a, part_b = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        buf_a = ops.pattern_fwd_blur(r_a, KF, sigma, s0, s1, ks, bs, True)
        buf_b = ops.pattern_fwd_blur(r_b, KF, sigma, s0, s1, ks, bs, True)
        for k in range(5):
            gtex_k = (rng.standard_normal((s1, s0)) * (k + 1)).astype(np.float32)
            gtex_k[rng.random((s1, s0)) < 0.3] = 0.0
            li = rng.standard_normal(n_slots).astype(np.float32)
            hdr = np.zeros(16, np.int32)
            if variant == "guard":
                d = 7 if k in (1, 3) else 0  # (steps 1 and 3 are NOT applied)
                hdr[0], hdr[1], hdr[2] = 4096 + d, 4096, d
            w = 0.1 if k != 2 else 0.0
            for acc in (acc_a, acc_b):
                acc.fill_(0.0)
                acc[0], acc[-1] = 7.0, 7.0
                acc[1:1 + T] = dev(gtex_k).reshape(-1)
                acc[1 + T:1 + T + n_slots] = dev(li)
                acc[1 + T + n_slots:-1].view(torch.int32).copy_(torch.from_numpy(hdr).cuda())
            gt_a, sl_a, hd_a = acc_a[1:1 + T].view(s1, s0), acc_a[1 + T:1 + T + n_slots], acc_a[1 + T + n_slots:-1].view(torch.uint8)
            gt_b, sl_b, hd_b = acc_b[1:1 + T].view(s1, s0), acc_b[1 + T:1 + T + n_slots], acc_b[1 + T + n_slots:-1].view(torch.uint8)
            g_a, g_b = torch.empty_like(r_a), torch.empty_like(r_b)
            dot_a = (a_d, b_d, part_a) if variant == "dot" else None
            dot_b = (a_d, b_d, part_b) if variant == "dot" else None
            # ---- the two launches
            aa = ops.adam_args(r_a, m_a, v_a, st_a, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=g_a, dot=dot_a,
                               guard=hd_a if variant == "guard" else None)
            gd_a, gr_a, val_a = ops.pattern_bwd_blur(r_a, KF, sigma, s0, s1, buf_a[1], buf_a[2], gt_a, w, buf_a[3], ks, bs, loss_in=None if dot_a else sl_a, loss_div=4.0,
                                                     adam=aa, scratch=torch.empty_like(buf_a[1]))
            if k == 0 and variant == "slots":
                gd_o, gr_o, val_o = oracle.pattern_bwd_blur(rays.copy(), KF, sigma, s0, s1, host(buf_a[1]), host(buf_a[2]), gtex_k, w, host(buf_a[3]), ks, bs, loss_in=li, loss_div=4.0)
            buf_a = ops.pattern_fwd_blur(r_a, KF, sigma, s0, s1, ks, bs, True, out=buf_a, zero=acc_a[1:-1])
            # ---- the one launch
            ab = ops.adam_args(r_b, m_b, v_b, st_b, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=g_b, dot=dot_b,
                               guard=hd_b if variant == "guard" else None)
            res = ops.pattern_step(r_b, KF, sigma, s0, s1, buf_b, gt_b, w, ks, bs, ab, acc_b[1:-1], sync, rays_kept=kept, check_kept=k > 0,
                                   loss_in=None if dot_b else sl_b, loss_div=4.0, epoch=k + 1)
            assert res is not None
            gd_b, gr_b, val_b = res
            what = f"{variant} step {k}"
            assert torch.equal(gd_a, gd_b) and (gr_a is None) == (gr_b is None) and (gr_a is None or torch.equal(gr_a, gr_b)), what
            assert torch.equal(val_a, val_b), (what, val_a, val_b)
            applied = not (variant == "guard" and k in (1, 3))
            if applied:
                assert torch.equal(g_a, g_b), what
            assert torch.equal(r_a, r_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b) and float(st_a) == float(st_b), what
            for x, y, nm in zip(buf_a, buf_b, ("pts", "tsum", "tsor", "ws", "tex")):
                assert torch.equal(x, y), (what, nm)
            assert torch.equal(acc_a, acc_b) and float(acc_b[0]) == 7.0 and float(acc_b[-1]) == 7.0 and float(acc_b[1:-1].abs().max()) == 0.0, what
            assert torch.equal(kept[(k + 1) & 1], r_b), what  # (launch k + 1 of this sync buffer: its parity's half)
            sw = sync.view(torch.int32).cpu().numpy()
            flags = sw[768::128]
            assert (sw[:8] == 0).all() and sw[256] == 0 and sw[512] == 0 and len(flags) == 64 and (flags == k + 1).all(), (what, sw[:8], flags)
            # (nothing stale, no time-out; the counters re-armed; the 64 flags at this launch's epoch)
            if variant == "guard":
                assert (sw[18:34] == hdr).all(), what
            if k == 0 and variant == "slots":
                np.testing.assert_allclose(host(gd_b), gd_o, rtol=3e-4, atol=3e-5 * max(np.abs(gd_o).max(), 1e-20))
                np.testing.assert_allclose(host(val_b), val_o, rtol=2e-5, atol=2e-5)
        steps = 3.0 if variant == "guard" else 5.0
        assert float(st_b) == steps
        if variant == "slots":
            # an edit behind torch's back between two steps: the launch that was told the texture is the kept pattern's finds out
            r_b[0, 0] += 1e-3
            acc_b.fill_(0.0)
            ab = ops.adam_args(r_b, m_b, v_b, st_b, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=4.0, grad_out=torch.empty_like(r_b))
            assert ops.pattern_step(r_b, KF, sigma, s0, s1, buf_b, acc_b[1:1 + T].view(s1, s0), 0.1, ks, bs, ab, acc_b[1:-1], sync, rays_kept=kept, check_kept=True,
                                    loss_in=acc_b[1 + T:1 + T + n_slots], loss_div=4.0, epoch=6) is not None
            assert int(sync.view(torch.int32)[4]) == 1
    # declined shapes: a footprint beyond the launch's LDS window, another kernel size
    r, m, v, st = fresh()
    aa = ops.adam_args(r, m, v, st, torch.zeros(1, dtype=torch.int32, device="cuda"), 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_out=torch.empty_like(r))
    bufs = ops.pattern_fwd_blur(r, KF, 400.0, s0, s1, 5, bs, True)
    z = torch.zeros(T + 16, device="cuda")
    sy = torch.zeros(35840, dtype=torch.uint8, device="cuda")
    kp = torch.zeros((2,) + tuple(r.shape), device="cuda")
    assert ops.pattern_step(r, KF, 400.0, s0, s1, bufs, z[:T].view(s1, s0), 0.1, 5, bs, aa, z, sy, rays_kept=kp) is None
    assert ops.pattern_step(r, KF, sigma, s0, s1, bufs, z[:T].view(s1, s0), 0.1, 3, bs, aa, z, sy, rays_kept=kp) is None
    assert torch.equal(r, dev(rays)) and float(st) == 0.0 and int(sy.view(torch.int32).abs().sum()) == 0  # (declined: nothing launched)


def test_dpp_three_chain_reduction_against_shuffle_reference(tmp_path):
    """the interleaved DPP / row_bcast wave reduction that make_widepk uses for the packet bounds
    (ffx_trace.hip: wave_reduce3_nn), compiled as the stand-alone checker tools/ubench/reduce3_check.hip and compared
    with a plain reduction on random data (4096 waves x 3 chains x min/max).  A wrong bound would make the packet
    test non-conservative, i.e. missed hits — which the parity tests above would only see where it happens."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "reduce3_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-w", "-o", exe, os.path.join(root, "tools", "ubench", "reduce3_check.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr


def test_wide_overlay_builders_give_identical_images(oracle, monkeypatch):
    """the 64-wide overlay only decides which boxes a packet visits, never what it hits: the three host builders
    (greedy SAH cut by area — the default —, by triangle count, and the first, layered one; FFX_WIDE_BUILD, read by
    ffx_bvh_build_host) must render bit-identical images and identical primitive ids — and the same as the oracle."""
    sc = scenes.vocalfold(width=72, height=64, tex=96, frames=3, n_fold=24, tube=(24, 32))
    sd = scene_desc.scene_desc(sc, shadows=True)
    tex = _tex(sc)
    cam = scene_desc.camera_from_sensor(sc.camera)
    imgs, prims, shapes_seen = [], [], []
    for mode in ("area", "count", "layers"):
        monkeypatch.setenv("FFX_WIDE_BUILD", mode)
        go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 2))
        shapes_seen.append((gd.info.n_wide, gd.info.wide_depth))
        imgs.append(gd.render_fwd(sd, dev(alb), tex, 8, seed=11))
        prims.append(gd.trace_primary(cam, 4, 1, 5)[2])
    assert len(set(shapes_seen)) > 1, shapes_seen  # they really are different trees
    for k in (1, 2):
        assert torch.equal(imgs[0], imgs[k]) and torch.equal(prims[0], prims[k])
    _, _, p_o = go.trace_primary(cam, 4, 1, 5)
    assert (host(prims[0]) == p_o).mean() > 0.9995


def test_sparse_adjoint_agrees_where_the_texture_is_not_zero(oracle):
    """FFX_RENDER_SPARSE_ADJOINT (include/ffx.h): with a sparse projector texture the cache-writing forward may skip
    the samples whose four bilinear taps are all exactly zero.  Same image bit for bit; the texture gradient agrees with
    the full one (and with the oracle's) at every texel whose value is not zero — the only texels a pattern optimiser's
    chain reads, since a zero texel has no splat within reach of the blur that produced it."""
    sc = scenes.vocalfold(width=72, height=64, tex=128, frames=3, n_fold=24, tube=(24, 32))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 8))
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    rng = np.random.default_rng(5)
    pts = (rng.random((12, 2)) * 0.8 + 0.1).astype(np.float32)  # a dozen dots on 128^2: most texels are exactly zero
    tex = ops.blur_fwd(ops.splat_fwd(dev(pts), 4.0, "sum", -1, 128, 128)).unsqueeze(-1).contiguous()
    zero = host(tex)[..., 0] == 0.0
    assert 0.5 < zero.mean() < 0.99
    gimg = dev(rng.standard_normal((64, 72, 3)).astype(np.float32))
    out = {}
    for sparse in (False, True):
        cache = torch.zeros(ops.render_cache_bytes(72, 64, 16), dtype=torch.uint8, device="cuda")
        img = gd.render_fwd(sd, dev(alb), tex, 16, seed=2, cache=cache, sparse_adjoint=sparse)
        out[sparse] = (img, host(gd.render_bwd_cached(sd, dev(alb), cache, 16, gimg))[..., 0])
    assert torch.equal(out[False][0], out[True][0])
    full, sp = out[False][1], out[True][1]
    scale = float(np.abs(full).max())
    assert scale > 0 and np.abs(full[zero]).max() > 0  # the full adjoint does reach dark texels ...
    np.testing.assert_allclose(sp[~zero], full[~zero], rtol=0, atol=1e-3 * scale)  # ... the sparse one agrees everywhere else
    g_o = go.render_bwd(sd, alb, 16, 2, host(gimg))[..., 0]
    np.testing.assert_allclose(sp[~zero], g_o[~zero], rtol=0, atol=2e-3 * scale)


@pytest.mark.parametrize("env", [{}, {"FFX_WIDE": "0"}, {"FFX_TRAVERSAL": "lane"}, {"FFX_BINS": "0"}])
@pytest.mark.parametrize("ch", [1, 3])
def test_principled_materials_match_the_oracle(oracle, env, ch, monkeypatch):
    """material rows (include/ffx.h FFX_MAT_*: Mitsuba's `principled` BSDF, reflection side) through every render entry
    point: forward, re-tracing adjoint, cache-writing forward + cached adjoint (second footprint), fp16 film; one shape
    stays Lambert.  The oracle's BSDF itself is pinned by tests/test_bruteforce_cpu.py."""
    from tests.test_bruteforce_cpu import material_rows

    for k in ("FFX_TRAVERSAL", "FFX_WIDE", "FFX_BINS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    xf = _rand_xforms(2, 5)
    go, gd, alb = _pair(oracle, sc, frame=2, xforms=xf)
    rng = np.random.default_rng(0)
    gimg = rng.standard_normal((44, 52, 3)).astype(np.float32)
    tex = _tex(sc, ch)
    for trial, fixed in enumerate(({}, {"anisotropic": 0.0, "clearcoat": 0.0, "sheen": 0.0, "flatness": 0.0, "metallic": 0.0, "spec_trans": 0.0, "spec_tint": 0.0})):
        mats = material_rows(2, 11 + trial, **fixed)
        if trial == 0:
            mats[0, 3] = 0.0  # a Lambert shape among principled ones
        for shadows, spp in ((True, 9), (False, 70)):
            sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=shadows, mat_stride=16)
            img_o = go.render_fwd(sd, mats, host(tex), spp, seed=3)
            img_d = host(gd.render_fwd(sd, dev(mats), tex, spp, seed=3))
            scale, _ = _assert_image_close(img_d, img_o, spp, frac=2e-4, rel=1e-4, what=f"{env} materials")
            assert scale > 0.02
            # not the Lambert image
            sd3 = scene_desc.scene_desc(sc, tex_channels=ch, shadows=shadows)
            assert np.abs(host(gd.render_fwd(sd3, dev(mats[:, :3].copy()), tex, spp, seed=3)) - img_d).max() > 0.02 * scale
            g_o = go.render_bwd(sd, mats, spp, 3, gimg)
            g_d = host(gd.render_bwd(sd, dev(mats), spp, 3, dev(gimg)))
            gs = float(np.abs(g_o).max())
            gerr = np.abs(g_d - g_o)
            assert gs > 0 and (gerr > 1e-3 * gs).mean() <= 1e-3 and gerr.max() <= 0.1 * gs
            # cache-writing forward: same image, cached adjoint = re-tracing adjoint
            nbytes = ops.render_cache_bytes_sd(sd, spp)
            assert nbytes > ops.render_cache_bytes(52, 44, spp) + 112 * 52 * 44 - 128
            cache = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            img_c = gd.render_fwd(sd, dev(mats), tex, spp, seed=3, cache=cache)
            torch.testing.assert_close(img_c, torch.from_numpy(img_d).cuda(), rtol=1e-4, atol=1e-5 * scale)
            hdr = host(cache[:12]).view(np.uint32)
            assert hdr[2] == 0 and hdr[0] <= hdr[1]
            g_c = host(gd.render_bwd_cached(sd, dev(mats), cache, spp, dev(gimg)))
            cerr = np.abs(g_c - g_o)
            assert (cerr > 1e-3 * gs).mean() <= 1e-3 and cerr.max() <= 0.1 * gs
        # fp16 film
        sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=16)
        h_d = host(gd.render_fwd(sd, dev(mats), tex, 9, seed=3, fp16=True)).astype(np.float32)
        h_o = go.render_fwd(sd, mats, host(tex), 9, seed=3, fp16=True).astype(np.float32)
        _assert_image_close(h_d, h_o, 9, frac=2e-3, rel=2e-3, what="fp16 materials")
    # a material table that does not match the scene description is refused
    with pytest.raises(ValueError):
        gd.render_fwd(scene_desc.scene_desc(sc, tex_channels=ch, mat_stride=16), dev(alb), tex, 4)


@pytest.mark.parametrize("env", [{}, {"FFX_WIDE": "0"}, {"FFX_TRAVERSAL": "lane"}, {"FFX_REFIT": "levels"}])
def test_interpolated_shading_normals_match_the_oracle(oracle, env, monkeypatch):
    """ffx_smooth (VERDICT r2 missing 2; fireflies/scene.py:243-251 -> Mitsuba re-derives vertex normals per update and
    shades in the interpolated frame): the update's vertex normals, the per-slot copies, and the render / both adjoints with
    a smooth and a flat shape in one scene, Lambert and principled rows, every walk and both refit paths, over several poses
    and animation frames — against the oracle, whose float64 cross-check is tests/test_bruteforce_cpu.py."""
    from tests.test_bruteforce_cpu import material_rows

    for k in ("FFX_TRAVERSAL", "FFX_WIDE", "FFX_REFIT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.vocalfold(width=56, height=48, tex=80, frames=4, n_fold=20, tube=(20, 24))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    rng = np.random.default_rng(3)
    gimg = rng.standard_normal((48, 56, 3)).astype(np.float32)
    tex = _tex(sc, 1)
    mats = material_rows(2, 17, anisotropic=0.0)
    for smooth in ([True, True], [False, True]):
        gd = ops.DeviceGeometry(pool, tris, shape, off, smooth=smooth)
        go = oracle.Geometry(pool, tris, shape, off, smooth=smooth)
        for it in range(3):
            xf = _rand_xforms(2, 60 + it)
            xf[1] = xf[1] @ np.diag([1.0, 1.0 + 0.3 * it, 1.0, 1.0]).astype(np.float32)  # a non-uniform scale: normals follow the deformed surface
            offs = (off + np.minimum(it, nfr - 1) * stride).astype(np.int32)
            gd.update(xf, offs)
            go.update(xf, offs)
            _ = gd.blob  # (orders the current stream behind the refit on the side stream)
            vn_d, vn_o = host(gd._smooth[1]["vn"]), go.vertex_normals
            np.testing.assert_allclose(vn_d, vn_o, rtol=0, atol=2e-6)
            # the per-slot copies next to the flagged records
            info = gd.info
            nrec = host(gd.blob[int(info.off_nrec): int(info.off_nrec) + 48 * int(info.n_tris)]).view(np.float32).reshape(-1, 3, 4)
            recs = host(gd.blob[int(info.off_recs): int(info.off_recs) + 48 * int(info.n_tris)]).view(np.float32).reshape(-1, 12)
            flagged = recs[:, 11] != 0
            prim, shp = recs[:, 9].view(np.int32), recs[:, 10].view(np.int32)
            assert (flagged == np.asarray(smooth)[shp]).all()
            vbase = gd._smooth[1]["vbase"]
            want = vn_d[(vbase[shp][:, None] + tris[prim])]
            np.testing.assert_array_equal(nrec[flagged][:, :, :3], want[flagged])
            for rows, stride_m in ((alb, 0), (mats, 16)):
                sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=stride_m)
                img_d = host(gd.render_fwd(sd, dev(rows), tex, 9, seed=5))
                img_o = go.render_fwd(sd, rows, host(tex), 9, seed=5)
                scale, _ = _assert_image_close(img_d, img_o, 9, frac=5e-4, rel=2e-4, what=f"{env} smooth={smooth} pose {it} stride {stride_m}")
                assert scale > 0.02
            g_o = go.render_bwd(sd, mats, 9, 5, gimg)
            g_d = host(gd.render_bwd(sd, dev(mats), 9, 5, dev(gimg)))
            gs = float(np.abs(g_o).max())
            gerr = np.abs(g_d - g_o)
            assert gs > 0 and (gerr > 1e-3 * gs).mean() <= 1e-3 and gerr.max() <= 0.1 * gs
            cache = torch.zeros(ops.render_cache_bytes_sd(sd, 9), dtype=torch.uint8, device="cuda")
            gd.render_fwd(sd, dev(mats), tex, 9, seed=5, cache=cache)
            g_c = host(gd.render_bwd_cached(sd, dev(mats), cache, 9, dev(gimg)))
            cerr = np.abs(g_c - g_o)
            assert (cerr > 1e-3 * gs).mean() <= 1e-3 and cerr.max() <= 0.1 * gs
    # ... and it is not the flat-shaded image
    gf = ops.DeviceGeometry(pool, tris, shape, off)
    gf.update(xf, offs)
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True)
    flat = host(gf.render_fwd(sd, dev(alb), tex, 9, seed=5))
    gs2 = ops.DeviceGeometry(pool, tris, shape, off, smooth=[True, True])
    gs2.update(xf, offs)
    assert np.abs(host(gs2.render_fwd(sd, dev(alb), tex, 9, seed=5)) - flat).max() > 0.02 * float(flat.max())
    # a flat geometry writes no flags: bit-identical to what it rendered before the feature existed (same kernels, same records)
    recs_f = host(gf.blob[int(gf.info.off_recs): int(gf.info.off_recs) + 48 * int(gf.info.n_tris)]).view(np.float32).reshape(-1, 12)
    assert (recs_f[:, 11] == 0).all()


@pytest.mark.parametrize("env", [{}, {"FFX_WIDE": "0"}, {"FFX_TRAVERSAL": "lane"}])
def test_textured_base_colour_matches_the_oracle(oracle, env, monkeypatch):
    """texture-valued base colours (`<mat>.brdf_0.base_color.data`, main.py:120-153; VERDICT r2 missing 3): material rows that
    select a texture (FFX_MAT_BASE_TEX), texture coordinates per leaf slot interpolated at the hit, repeat wrap, bilinear —
    a principled and a Lambert row on two textures of odd sizes, coordinates outside [0,1], together with interpolated
    normals, forward and re-tracing adjoint, packet and lane kernels; the footprint cache refuses such a scene."""
    from tests.test_bruteforce_cpu import material_rows

    for k in ("FFX_TRAVERSAL", "FFX_WIDE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.vocalfold(width=56, height=48, tex=80, frames=3, n_fold=20, tube=(20, 24))
    rng = np.random.default_rng(4)
    for m in sc.meshes:
        m.uv = (rng.random((m.frames.shape[1], 2)) * 3.0 - 1.0).astype(np.float32)
    btex = [rng.random((17, 23, 3)).astype(np.float32), rng.random((8, 5, 3)).astype(np.float32)]
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    xf = _rand_xforms(2, 7)
    offs = (off + stride).astype(np.int32)
    for smooth in (None, [True, False]):
        gd = ops.DeviceGeometry(pool, tris, shape, off, smooth=smooth)
        go = oracle.Geometry(pool, tris, shape, off, smooth=smooth)
        gd.update(xf, offs)
        go.update(xf, offs)
        order_d = host(gd.blob[int(gd.info.off_order): int(gd.info.off_order) + 4 * int(gd.info.n_tris)]).view(np.int32)
        order_o = go.blob[go.info.off_order: go.info.off_order + 4 * go.info.n_tris].view(np.int32)
        suv_d, suv_o = dev(scenes.slot_uv_table(order_d, tris, shape, sc.meshes)), scenes.slot_uv_table(order_o, tris, shape, sc.meshes)
        btex_d = [dev(b) for b in btex]
        mats = material_rows(2, 23, anisotropic=0.0)
        mats[0, 15], mats[1, 15], mats[1, 3] = 1.0, 2.0, 0.0  # a principled row on texture 0, a Lambert row on texture 1
        tex = _tex(sc, 1)
        gimg = rng.standard_normal((48, 56, 3)).astype(np.float32)
        kw = dict(tex_channels=1, shadows=True, mat_stride=16)
        sd_d = scene_desc.scene_desc(sc, base_tex=[(b.data_ptr(), b.shape[1], b.shape[0]) for b in btex_d], slot_uv=suv_d.data_ptr(), **kw)
        sd_o = scene_desc.scene_desc(sc, base_tex=[(b.ctypes.data, b.shape[1], b.shape[0]) for b in btex], slot_uv=suv_o.ctypes.data, **kw)
        for spp in (9, 70):
            img_d = host(gd.render_fwd(sd_d, dev(mats), tex, spp, seed=6))
            img_o = go.render_fwd(sd_o, mats, host(tex), spp, seed=6)
            scale, _ = _assert_image_close(img_d, img_o, spp, frac=5e-4, rel=2e-4, what=f"{env} smooth={smooth} spp {spp}")
            assert scale > 0.02
        g_d = host(gd.render_bwd(sd_d, dev(mats), 9, 6, dev(gimg)))
        g_o = go.render_bwd(sd_o, mats, 9, 6, gimg)
        gs = float(np.abs(g_o).max())
        gerr = np.abs(g_d - g_o)
        assert gs > 0 and (gerr > 1e-3 * gs).mean() <= 1e-3 and gerr.max() <= 0.1 * gs
        # the textures matter, and a row without one is untouched by their presence
        plain = mats.copy()
        plain[:, 15] = 0.0
        sd_p = scene_desc.scene_desc(sc, **kw)
        img_p = host(gd.render_fwd(sd_p, dev(plain), tex, 9, seed=6))
        img_t = host(gd.render_fwd(sd_d, dev(mats), tex, 9, seed=6))
        assert np.abs(img_p - img_t).max() > 0.05 * float(img_p.max())
        np.testing.assert_allclose(host(gd.render_fwd(sd_d, dev(plain), tex, 9, seed=6)), img_p, rtol=1e-5, atol=1e-6 * float(img_p.max()))
        # the footprint cache folds one base colour per shape: refused for textured scenes, through the C ABI and by the wrapper
        from fireflies_amd import functional as Fn

        assert not Fn.cache_supported(sd_d, 9)
        cache = torch.zeros(ops.render_cache_bytes_sd(sd_d, 9), dtype=torch.uint8, device="cuda")
        with pytest.raises(Exception, match="textured base colours"):
            gd.render_fwd(sd_d, dev(mats), tex, 9, seed=6, cache=cache)


@pytest.mark.parametrize("env", [{}, {"FFX_TRAVERSAL": "lane"}])
def test_material_table_inside_the_scene_description(oracle, env, monkeypatch):
    """ffx_scene_desc.mat_h (round 3): up to 128 floats of material table travel as a KERNEL ARGUMENT — the render calls take no
    material pointer, a randomisation enqueues no upload.  Same bits as the device table for [S,3] albedos and [S,16] rows,
    forward, cache-writing forward, cached and re-tracing adjoint; the oracle honours the field too; a table that does not fit
    or does not match n_shapes x stride is refused."""
    from tests.test_bruteforce_cpu import material_rows

    for k in ("FFX_TRAVERSAL",):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 9))
    tex = _tex(sc, 1)
    gimg = dev(np.random.default_rng(5).standard_normal((44, 52, 3)).astype(np.float32))
    for rows, stride in ((alb, 0), (material_rows(2, 31), 16)):
        sd_dev = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=stride)
        sd_host = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=stride, host_mats=rows)
        assert sd_dev.n_mat_h == 0 and sd_host.n_mat_h == rows.size
        a = gd.render_fwd(sd_dev, dev(rows), tex, 9, seed=4)
        b = gd.render_fwd(sd_host, None, tex, 9, seed=4)
        assert torch.equal(a, b)
        assert torch.equal(b, gd.render_fwd(sd_host, dev(np.zeros_like(rows)), tex, 9, seed=4))  # a pointer that is passed anyway is ignored
        np.testing.assert_array_equal(go.render_fwd(sd_host, np.zeros_like(rows), host(tex), 9, seed=4), go.render_fwd(sd_dev, rows, host(tex), 9, seed=4))
        ga = gd.render_bwd(sd_dev, dev(rows), 9, 4, gimg)
        gb = gd.render_bwd(sd_host, None, 9, 4, gimg)
        gs = float(ga.abs().max())
        assert float((ga - gb).abs().max()) <= 1e-3 * gs  # (float atomics: order only)
        cache = torch.zeros(ops.render_cache_bytes_sd(sd_host, 9), dtype=torch.uint8, device="cuda")
        assert torch.equal(gd.render_fwd(sd_host, None, tex, 9, seed=4, cache=cache), gd.render_fwd(sd_dev, dev(rows), tex, 9, seed=4, cache=torch.zeros_like(cache)))
        gc = gd.render_bwd_cached(sd_host, None, cache, 9, gimg)
        assert float((gc - ga).abs().max()) <= 2e-3 * gs
    with pytest.raises(ValueError):  # no table at all
        gd.render_fwd(scene_desc.scene_desc(sc, tex_channels=1), None, tex, 4)
    with pytest.raises(ValueError):  # rows of the wrong stride
        scene_desc.scene_desc(sc, tex_channels=1, mat_stride=16, host_mats=alb)
    bad = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, host_mats=alb)
    bad.n_mat_h = 5  # not n_shapes x stride: refused by the library
    with pytest.raises(Exception, match="scene description"):
        gd.render_fwd(bad, None, tex, 4)
    many = scene_desc.scene_desc(sc, tex_channels=1)
    many.n_shapes = 50
    assert scene_desc.set_host_materials(many, np.zeros((50, 3), np.float32)) is False and many.n_mat_h == 0  # 150 floats: stays a device table


def test_principled_materials_mid_size_and_abi_errors(oracle):
    """material rows at 256x256x64 spp on the full-detail vocal fold (one pixel per wave, the production launch shape) against
    the oracle, with the reference's vocal-fold randomisation (specular 0 .. 0.75, roughness 0.5); and the C ABI's refusals:
    an unknown material stride, material rows that are not 16-byte aligned."""
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.vocalfold(width=256, height=256, tex=500, frames=4)
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 3))
    tex = _tex(sc, 1)
    mats = material_rows(2, 5, anisotropic=0.0, clearcoat=0.0, sheen=0.0, flatness=0.0, metallic=0.0, spec_trans=0.0, spec_tint=0.0, roughness=0.5)
    mats[:, 8] = [2.0 / (1.0 - np.sqrt(0.08 * s_)) - 1.0 for s_ in (0.0, 0.75)]  # specular 0 (eta 1: no lobe) and 0.75
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16)
    img_d = host(gd.render_fwd(sd, dev(mats), tex, 64, seed=12))
    img_o = go.render_fwd(sd, mats, host(tex), 64, seed=12)
    scale, err = _assert_image_close(img_d, img_o, 64, frac=2e-4, rel=1e-4, what="materials 256x256x64")
    assert scale > 0.05 and float(err.mean()) < 2e-6 * scale
    # ---- refusals
    from fireflies_amd import _lib

    a = _lib.api()
    img = torch.empty((256, 256, 3), device="cuda")
    bad = scene_desc.scene_desc(sc, tex_channels=1, mat_stride=5)
    rc = a.lib.ffx_render_fwd(gd.blob.data_ptr(), C.byref(gd.info), C.byref(bad), dev(mats).data_ptr(), tex.data_ptr(), 4, 0, 0, img.data_ptr(), None)
    assert rc == -1 and b"scene description" in a.lib.ffx_last_error()
    shifted = torch.zeros(2 * 16 + 1, device="cuda")[1:]  # 4-byte aligned only
    rc = a.lib.ffx_render_fwd(gd.blob.data_ptr(), C.byref(gd.info), C.byref(sd), shifted.data_ptr(), tex.data_ptr(), 4, 0, 0, img.data_ptr(), None)
    assert rc == -1 and b"16-byte aligned" in a.lib.ffx_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("ch", [1, 3])
def test_material_rows_forward_and_adjoint_are_transposes(ch):
    """independent of the oracle: with material rows the render is still linear in the texture, and both adjoints (cached,
    re-tracing) are its transpose — <render(tex + d) - render(tex), g> = <d, adjoint(g)> for random d, g (float64 sums)."""
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.vocalfold(width=64, height=48, tex=72, frames=3, n_fold=20, tube=(20, 24))
    S = len(sc.meshes)
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    gd.update(_rand_xforms(S, 4))
    mats = dev(material_rows(S, 21))
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=16)
    g = torch.Generator(device="cuda").manual_seed(5)
    tex = torch.rand((72, 72, ch), device="cuda", generator=g)
    d = torch.randn((72, 72, ch), device="cuda", generator=g) * 0.1
    gimg = torch.randn((48, 64, 3), device="cuda", generator=g)
    spp = 12
    base = gd.render_fwd(sd, mats, tex, spp, seed=8).double()
    pert = gd.render_fwd(sd, mats, tex + d, spp, seed=8).double()
    lhs = float(((pert - base) * gimg.double()).sum())
    g_re = gd.render_bwd(sd, mats, spp, 8, gimg).double()
    cache = torch.zeros(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
    gd.render_fwd(sd, mats, tex, spp, seed=8, cache=cache)
    g_ca = gd.render_bwd_cached(sd, mats, cache, spp, gimg).double()
    for name, gt in (("re-tracing", g_re), ("cached", g_ca)):
        rhs = float((d.double() * gt).sum())
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs)), (name, lhs, rhs)
    assert abs(lhs) > 1e-3
    # linearity itself: render(2 tex) - render(tex) = render(tex) - render(0)
    zero = gd.render_fwd(sd, mats, torch.zeros_like(tex), spp, seed=8).double()
    two = gd.render_fwd(sd, mats, 2 * tex, spp, seed=8).double()
    torch.testing.assert_close(two - base, base - zero, rtol=1e-4, atol=1e-5 * float(base.max()))


def test_the_in_kernel_update_is_skipped_when_the_adjoint_cache_dropped_samples(oracle):
    """ffx_adam_args.guard (round 4, advisor): the Adam + clamp_to_fov update that rides on ffx_pattern_bwd_blur looks at the adjoint
    cache's header — `dropped` != 0 means K9 has poisoned the gradient with NaN — and then leaves rays, both moments and the step count
    untouched, so that the caller can repeat the step with the re-tracing adjoint instead of finding NaN in its optimiser state.
    HIP and oracle; the same launch with a clean header applies the update."""
    rng = np.random.default_rng(3)
    n, (s0, s1), sigma, ks, bs = 32, (64, 48), 10.0, 5, 3.0
    g2 = load_golden("g2_projection.npz")
    KF = (g2["K"] @ FLIP_Y).astype(np.float32)
    KFi = np.linalg.inv(KF.astype(np.float64)).astype(np.float32)
    ndc = (rng.random((n, 3)) * np.array([0.8, 0.8, 0.0]) + np.array([0.1, 0.1, -1.0])).astype(np.float32)
    rays = oracle.transform_points(ndc, KFi)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    gtex = rng.standard_normal((s1, s0)).astype(np.float32)
    gtex[0, 0] = np.nan  # what K9 leaves behind when the cache dropped samples
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    for dropped in (7, 0):
        hdr = np.zeros(16, np.uint32)
        hdr[0], hdr[1], hdr[2] = 4096 + dropped, 4096, dropped
        guard = torch.from_numpy(hdr.view(np.uint8).copy()).cuda()
        r, m, v, st = dev(rays).clone(), torch.zeros(n, 3, device="cuda"), torch.zeros(n, 3, device="cuda"), torch.zeros((), device="cuda")
        _, ts, to, ws, _ = ops.pattern_fwd_blur(r, KF, sigma, s0, s1, ks, bs, True)
        aa = ops.adam_args(r, m, v, st, counter, 5e-3, 0.9, 0.999, 1e-8, KFi, 0.05, 0.95, 2, grad_div=1.0, grad_out=torch.empty_like(r), guard=guard)
        g_in = dev(gtex if dropped else np.nan_to_num(gtex))
        ops.pattern_bwd_blur(r, KF, sigma, s0, s1, ts, to, g_in, 0.1, ws, ks, bs, adam=aa, scratch=torch.empty_like(ts))
        if dropped:
            assert torch.equal(r, dev(rays)) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0 and float(st) == 0.0 and int(counter) == 0
        else:
            assert not torch.equal(r, dev(rays)) and float(st) == 1.0 and bool(torch.isfinite(r).all()) and int(counter) == 0


# ------------------------------------------------------------------ reconstruction filter (ffx_scene_desc.rfilter)
def test_filtered_film_adjoint_cache_is_an_arena_at_config_5_size(oracle):
    """Round-5 review, missing 4 / item 7: the filtered film's adjoint cache was a dense [pixel][sample] array — 5.4 GB of address space at
    BASELINE configs[4] (1024 x 1024 x 256 spp, material rows) for records of the few per cent of the pixels that hold a lit sample.  Now an arena
    of 64-sample blocks: 1.34 GB there (a quarter of the passes: configs[4]'s own 1 024-point pattern lights 17 % of the film).  The colon at 96 x 96 against the ORACLE's filtered adjoint, and at full size — 1024 x 1024 x 256, a dot
    pattern, FFX_RENDER_SPARSE_ADJOINT — against the library's own re-traced filtered adjoint; nothing dropped, a plausible share of the arena used."""
    from fireflies_amd import _abi
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.colon(width=96, height=96, tex=128, n_around=48, n_along=160)
    go, gd, alb = _pair(oracle, sc, frame=0, xforms=_rand_xforms(1, 3))
    mats = material_rows(1, 6)
    spp = 70
    sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=_abi.MAT_STRIDE, rfilter="gaussian")
    tex = _tex(sc, 1)
    rng = np.random.default_rng(4)
    gimg = rng.standard_normal((96, 96, 3)).astype(np.float32)
    cache = torch.empty(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device="cuda")
    gd.render_fwd(sd, dev(mats), tex, spp, seed=21, cache=cache)
    assert ops.render_cache_status(cache)[2] == 0
    gt_c = host(gd.render_bwd_cached(sd, dev(mats), cache, spp, dev(gimg), seed=21))
    gt_o = go.render_bwd(sd, mats, spp, 21, gimg)
    gs = float(np.abs(gt_o).max())
    ec = np.abs(gt_c - gt_o)
    assert gs > 0 and (ec > 1e-3 * gs).mean() <= 2e-3 and ec.max() <= 0.1 * gs, ((ec > 1e-3 * gs).mean(), ec.max() / gs)
    del gd, go
    # ---- configs[4] at full size: the workload's own scene, pattern and material randomisation
    from fireflies_amd import mi, workloads

    wl = workloads.colon(device="cuda", entity_device="cpu")
    with torch.no_grad():
        t = workloads.build_texture(wl).contiguous()
    torch.manual_seed(5)
    random.seed(5)
    wl.ff_scene.randomize()
    wl.mi_scene.rfilter = "gaussian"
    sdb = wl.mi_scene.scene_desc(tex_channels=1)
    nb = ops.render_cache_bytes_sd(sdb, 256)
    assert nb <= 1.4e9, nb  # (5.4e9 as a dense array)
    cache = torch.empty(nb, dtype=torch.uint8, device="cuda")
    g = wl.mi_scene.geom
    matsb = wl.mi_scene.materials_arg(sdb)
    tex3 = t.unsqueeze(-1).contiguous()
    img = g.render_fwd(sdb, matsb, tex3, 256, seed=2, cache=cache, sparse_adjoint=True)
    used, cap, dropped = ops.render_cache_status(cache)
    assert dropped == 0 and 0 < used <= cap and cap == (1024 * 1024 * 4) // 4 and bool(torch.isfinite(img).all())
    gb = torch.randn((1024, 1024, 3), device="cuda").sign_() / (3.0 * 1024 * 1024)
    a = g.render_bwd_cached(sdb, matsb, cache, 256, gb, seed=2)
    b = g.render_bwd(sdb, matsb, 256, 2, gb)
    nz = tex3 != 0  # (sparse records: gradients where the texture is not zero)
    sb = float(b[nz].abs().max())
    eb = (a[nz] - b[nz]).abs()
    assert sb > 0 and float((eb > 1e-3 * sb).float().mean()) <= 2e-3 and float(eb.max()) <= 0.1 * sb
    # the caller's answer to an overflow (the bench's loop reaches poses on which this pattern lights 26 % of the block-passes): the DENSE layout,
    # FFX_SHADOWS_CACHE_DENSE in the description — a block for every pass of every pixel, handed out without atomics; same records, same gradient
    del cache
    wl.mi_scene.set_cache_dense(True)
    sdd = wl.mi_scene.scene_desc(tex_channels=1)
    nd = ops.render_cache_bytes_sd(sdd, 256)
    assert int(sdd.shadows) & 4 and 5.0e9 <= nd <= 5.6e9, (int(sdd.shadows), nd)
    cache = torch.empty(nd, dtype=torch.uint8, device="cuda")
    img_d = g.render_fwd(sdd, matsb, tex3, 256, seed=2, cache=cache, sparse_adjoint=True)
    assert torch.equal(img_d, img) and ops.render_cache_status(cache) == (0, 1024 * 1024 * 4, 0)
    a_d = g.render_bwd_cached(sdd, matsb, cache, 256, gb, seed=2)
    ed = (a_d[nz] - b[nz]).abs()
    assert float((ed > 1e-3 * sb).float().mean()) <= 2e-3 and float(ed.max()) <= 0.1 * sb


@pytest.mark.parametrize("ch,rows,spp,stddev", [(1, "albedo", 8, 0.5), (3, "albedo", 8, 0.5), (1, "material_rows", 70, 0.5), (1, "albedo", 5, 0.3)])
def test_gaussian_reconstruction_filter_forward_and_adjoint_match_the_oracle(oracle, ch, rows, spp, stddev, monkeypatch):
    """hdrfilm's default filter, which every scene the reference loads gets: ffx_render_fwd_filtered / ffx_render_bwd_filtered against the
    oracle's (itself checked against the float64 brute force, tests/test_bruteforce_cpu.py) — image, fp16 film, texture gradient — for both
    texture layouts, Lambert and material rows, one and two 64-sample passes per pixel and a narrower filter; every other render entry point
    refuses the scene instead of rendering a box."""
    from fireflies_amd import _abi

    sc = scenes.vocalfold(width=72, height=64, tex=96, frames=3, n_fold=24, tube=(24, 32))
    go, gd, alb = _pair(oracle, sc, frame=1, xforms=_rand_xforms(2, 2))
    if rows == "material_rows":
        from tests.test_bruteforce_cpu import material_rows

        alb = material_rows(len(sc.meshes), 5)
    sd = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=_abi.MAT_STRIDE if rows == "material_rows" else 0, rfilter=("gaussian", stddev))
    assert sd.rfilter == _abi.RFILTER_GAUSSIAN
    tex = _tex(sc, ch)
    img_d = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=11))
    img_o = go.render_fwd(sd, alb, host(tex), spp, seed=11)
    assert img_o.max() > 0.05
    # (a sample that flips on an edge is spread over its window: the share of touched pixels is up to 25x the box render's)
    scale, _ = _assert_image_close(img_d, img_o, spp, frac=0.02, what=f"gaussian ch={ch} {rows} spp={spp}")
    sd_box = scene_desc.scene_desc(sc, tex_channels=ch, shadows=True, mat_stride=sd.mat_stride)
    box = host(gd.render_fwd(sd_box, dev(alb), tex, spp, seed=11))
    assert np.abs(box - img_d).max() > 0.02 * scale  # the field is read
    assert abs(float(box.mean()) - float(img_d.mean())) < 0.02 * float(box.mean())  # ... and the filter is normalised
    assert torch.equal(gd.render_fwd(sd, dev(alb), tex, spp, seed=11), gd.render_fwd(sd, dev(alb), tex, spp, seed=11))  # no atomics
    img_h = host(gd.render_fwd(sd, dev(alb), tex, spp, seed=11, fp16=True)).astype(np.float32)
    np.testing.assert_allclose(img_h, img_d, rtol=1e-3, atol=1e-4 * scale)
    # ---- adjoint
    rng = np.random.default_rng(1)
    gimg = rng.standard_normal((64, 72, 3)).astype(np.float32)
    gimg[:, :20] = 0.0  # (pixels whose whole window carries no gradient are skipped: the border of this block must still get its share)
    gt_d = host(gd.render_bwd(sd, dev(alb), spp, 11, dev(gimg)))
    gt_o = go.render_bwd(sd, alb, spp, 11, gimg)
    gs = float(np.abs(gt_o).max())
    assert gs > 0
    err = np.abs(gt_d - gt_o)
    assert (err > 1e-3 * gs).mean() <= 2e-3, f"{(err > 1e-3 * gs).mean():.2e}"
    assert err.max() <= 0.1 * gs
    # the two directions are transposes of each other on the device itself
    base = host(gd.render_fwd(sd, dev(alb), torch.zeros_like(tex), spp, seed=11)).astype(np.float64)
    lhs = float(((img_d - base) * gimg).sum())
    rhs = float((host(tex).astype(np.float64).reshape(gt_d.shape) * gt_d).sum())
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    # ---- store instead of re-trace (ABI 7): the filtered forward that also writes one record per sample of the lit pixels, and the adjoint
    # from those records — same image (bitwise), the re-traced adjoint's gradient up to the order of the float atomics, the oracle's own
    # cached pair (one record per sample of every pixel), unaffected by a re-fit in between, accumulating into the caller's buffer
    monkeypatch.delenv("FFX_RFC_CAP", raising=False)
    nb = ops.render_cache_bytes_sd(sd, spp)
    npx, up = 72 * 64, (lambda v: ((v + 127) // 128) * 128)
    # (round 6: the records live in an arena of 64-sample blocks — a block per pass of every pixel while that stays below 2^18 blocks, as here)
    blocks = npx * ((spp + 63) // 64)
    assert nb == up(up(up(64 + 8 * npx) + 4 * npx) + 16 * 64 * blocks) + (up(4 * 64 * blocks) if rows == "material_rows" else 0)
    cache = torch.full((nb,), 0xAB, dtype=torch.uint8, device="cuda")  # (garbage: every byte the adjoint reads is written by the forward)
    img_c = gd.render_fwd(sd, dev(alb), tex, spp, seed=11, cache=cache)
    assert torch.equal(img_c.cpu(), torch.from_numpy(img_d))
    # (an arena that holds a block for every pass of every pixel hands each pixel its own: nothing is taken from a counter, nothing can run out)
    assert ops.render_cache_status(cache) == (0, blocks, 0)
    pose = gd._vert_off_host.copy()
    gd.update(_rand_xforms(2, 99))  # re-fit to another pose: the cached adjoint needs neither the tree nor the camera
    acc = torch.full((sd.proj.tex_h, sd.proj.tex_w, ch), 1.5, device="cuda")
    assert gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), out=acc, seed=11) is acc
    gt_c = host(acc) - 1.5
    gd.update(_rand_xforms(2, 2), pose)
    ec = np.abs(gt_c - gt_o)
    assert (ec > 1e-3 * gs).mean() <= 2e-3 and ec.max() <= 0.1 * gs, ((ec > 1e-3 * gs).mean(), ec.max() / gs)
    np.testing.assert_allclose(gt_c, gt_d, rtol=0, atol=1e-3 * gs)
    img_oc, cache_o = go.render_fwd_cache(sd, alb, host(tex), spp, seed=11)
    np.testing.assert_array_equal(img_oc, img_o)
    np.testing.assert_allclose(go.render_bwd_cached(sd, alb, cache_o, spp, gimg, seed=11), gt_o, rtol=0, atol=1e-6 * gs)  # (the oracle's pair = its re-trace)
    with pytest.raises(ValueError, match="seed"):
        gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg))
    # sparse records (FFX_RENDER_SPARSE_ADJOINT): the same gradient wherever the texture is not zero
    tex_s = tex.clone()
    tex_s[: tex_s.shape[0] // 2] = 0.0
    gd.render_fwd(sd, dev(alb), tex_s, spp, seed=11, cache=cache, sparse_adjoint=True)
    gt_sp = host(gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), seed=11))
    nzs = host(tex_s).reshape(gt_d.shape) != 0
    np.testing.assert_allclose(gt_sp[nzs], gt_d[nzs], rtol=0, atol=1e-3 * gs)
    # an arena too small for the lit pixels (FFX_RFC_CAP: the overflow path): the image is the same, the pixels that found no block are counted,
    # the adjoint from such a cache is poisoned rather than silently incomplete — and the next plain forward starts from a clean header
    monkeypatch.setenv("FFX_RFC_CAP", "5")
    img_x = gd.render_fwd(sd, dev(alb), tex, spp, seed=11, cache=cache)
    used_x, cap_x, dropped_x = ops.render_cache_status(cache)
    assert torch.equal(img_x.cpu(), torch.from_numpy(img_d)) and cap_x == 5 and dropped_x > 0 and used_x > cap_x
    assert bool(torch.isnan(gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), seed=11).reshape(-1)[0]))
    monkeypatch.delenv("FFX_RFC_CAP")
    gd.render_fwd(sd, dev(alb), tex, spp, seed=11, cache=cache)
    assert ops.render_cache_status(cache)[2] == 0
    np.testing.assert_allclose(host(gd.render_bwd_cached(sd, dev(alb), cache, spp, dev(gimg), seed=11)), gt_d, rtol=0, atol=1e-3 * gs)
    if ch == 1:
        # forward + adjoint of a loss that is linear in the image in ONE render launch (ffx_render_fwd_adjoint_filtered): the same image, the same
        # gradient as the pair above (and the oracle's composition), also with the sparse flag where the texture is not zero
        img_f, gt_f = gd.render_fwd_adjoint(sd, dev(alb), tex, spp, 11, dev(gimg))
        assert torch.equal(img_f.cpu(), torch.from_numpy(img_d))
        ef = np.abs(host(gt_f) - gt_o)
        assert (ef > 1e-3 * gs).mean() <= 2e-3 and ef.max() <= 0.1 * gs
        np.testing.assert_allclose(host(gt_f), gt_d, rtol=0, atol=2e-4 * gs)  # (the order of the float atomics)
        _, gt_s = gd.render_fwd_adjoint(sd, dev(alb), tex, spp, 11, dev(gimg), sparse_adjoint=True)
        nz = host(tex).reshape(gt_d.shape) != 0
        np.testing.assert_allclose(host(gt_s)[nz], gt_d[nz], rtol=0, atol=2e-4 * gs)
        img_o2, gt_o2, _ = go.render_fwd_adjoint(sd, alb, host(tex), spp, 11, gimg)
        np.testing.assert_array_equal(img_o2, img_o)
        np.testing.assert_allclose(gt_o2, gt_o, rtol=0, atol=1e-6 * gs)
    else:
        with pytest.raises(ValueError, match="1-channel"):
            gd.render_fwd_adjoint(sd, dev(alb), tex, spp, 11, dev(gimg))
    lib, UNSUPPORTED = ops.api(), -3  # FFX_ERR_UNSUPPORTED
    img_t = torch.empty((64, 72, 3), device="cuda")
    mats_arg = dev(alb).data_ptr()
    rc = lib.lib.ffx_render_fwd(gd.blob.data_ptr(), C.byref(gd.info), C.byref(sd), mats_arg, tex.data_ptr(), spp, 11, 0, img_t.data_ptr(), None)
    assert rc == UNSUPPORTED and "reconstruction filter" in lib.lib.ffx_last_error().decode()
    rc = lib.lib.ffx_render_bwd(gd.blob.data_ptr(), C.byref(gd.info), C.byref(sd), mats_arg, spp, 11, 0, dev(gimg).data_ptr(), torch.zeros_like(tex).data_ptr(), None)
    assert rc == UNSUPPORTED


def test_gaussian_filter_through_the_python_api_and_autograd(oracle):
    """mi.Scene.rfilter = "gaussian": mi.render goes through the filtered entry points (also beside the previous render on the scene's second
    stream), autograd differentiates through the filtered film's adjoint cache (ABI 7; re-traced until round 5), and the optimiser takes the
    fused launch for its linear loss and the cache + K9 pair for a non-linear one."""
    from fireflies_amd import functional as Fn, mi, workloads
    from fireflies_amd.optim import PatternOptimizer

    wl = workloads.vocalfold(device="cuda", width=64, height=56, tex=96, grid=6, frames=5, n_fold=20, tube=(20, 24))
    ms = wl.mi_scene
    assert ms.rfilter == "box"
    tex = ms._params["tex.data"]
    tex = (tex.t if hasattr(tex, "t") else tex).detach().clone()
    box = mi.render(ms, spp=8, seed=3).torch().clone()
    ms.rfilter = "gaussian"
    sd = ms.scene_desc(tex_channels=1 if tex.dim() == 2 else int(tex.shape[-1]))
    assert sd.rfilter == 1 and Fn.cache_supported(sd, 8) and not Fn.cache_supported(sd, 2048)  # (per-sample records: up to 16 passes of 64 samples)
    a = mi.render(ms, spp=8, seed=3).torch().clone()
    b = mi.render(ms, spp=8, seed=3).torch().clone()  # (the second call runs on the other render stream)
    assert torch.equal(a, b) and not torch.equal(a, box)
    t = tex.clone().requires_grad_(True)
    ms._params["tex.data"] = t
    img = mi.render(ms, spp=8, seed=3).torch()
    w = torch.randn_like(img)
    (img * w).sum().backward()
    want = ms.geom.render_bwd(sd, ms.materials_arg(sd), 8, 3, w.contiguous())
    torch.testing.assert_close(t.grad.reshape(want.shape), want, rtol=1e-3, atol=1e-3 * float(want.abs().max()))
    ms._params["tex.data"] = tex
    # the coverage loss is linear in the image: under the gaussian film the step takes the record-writing forward + the adjoint from the records
    # (round 5: faster than the fused filtered launch, which FFX_FUSED_ADJOINT_FILTERED=1 still selects) — the same loss, the same update
    before = wl.laser._rays.detach().clone()
    lin = {}
    for fused_env in ("0", "1"):
        os.environ["FFX_FUSED_ADJOINT_FILTERED"] = fused_env
        try:
            wl.laser._rays = before.clone()
            opt = PatternOptimizer(ms, wl.ff_scene, wl.laser, sigma=10.0, tex_size=(96, 96), spp=4, lr=5e-3, reg_weight=0.1, base_seed=21, samples_per_step=2)
            res = opt.step()
        finally:
            os.environ.pop("FFX_FUSED_ADJOINT_FILTERED", None)
        assert opt.step_paths == ({"fused": 0, "cache_k9": 2, "retrace": 0} if fused_env == "0" else {"fused": 2, "cache_k9": 0, "retrace": 0})
        assert np.isfinite(float(res["loss"])) and not torch.equal(before, wl.laser._rays.detach())
        lin[fused_env] = (float(res["loss"]), wl.laser._rays.grad.detach().clone())
    assert lin["0"][0] == pytest.approx(lin["1"][0], rel=1e-5)
    assert float((lin["0"][1] - lin["1"][1]).abs().max()) <= 2e-3 * float(lin["1"][1].abs().max())
    # ... and a loss that is not linear in the image — the reference's own L1 (rasterization.py:579) — takes the filtered cache + its adjoint
    # (round 5; it re-traced before), with the same update as the re-tracing step
    from fireflies_amd.optim import image_l1_loss

    rays0 = wl.laser._rays.detach().clone()
    runs = {}
    for limit in ("32", "0"):  # (FFX_CACHE_LIMIT_GB=0: no cache fits — the re-tracing adjoint)
        Fn.CACHE_LIMIT_BYTES = int(float(limit) * (1 << 30))
        wl.laser._rays = rays0.clone()
        opt2 = PatternOptimizer(ms, wl.ff_scene, wl.laser, sigma=10.0, tex_size=(96, 96), spp=4, lr=5e-3, reg_weight=0.1, base_seed=21, samples_per_step=2,
                                loss_fn=image_l1_loss(a.float()))
        res2 = opt2.step()
        assert opt2.step_paths["cache_k9" if limit == "32" else "retrace"] == 2 and np.isfinite(float(res2["loss"]))
        runs[limit] = (float(res2["loss"]), wl.laser._rays.grad.detach().clone())
    Fn.CACHE_LIMIT_BYTES = 32 << 30
    assert runs["32"][0] == pytest.approx(runs["0"][0], rel=1e-5)
    gmax = float(runs["0"][1].abs().max())
    assert gmax > 0 and float((runs["32"][1] - runs["0"][1]).abs().max()) <= 2e-3 * gmax


def test_deterministic_adjoint_is_bitwise_reproducible_and_cross_checks_the_atomic_paths(oracle, monkeypatch):
    """ffx_render_bwd_det (ABI 7; SURVEY 5 "race detection", 7.4 "deterministic mode"): the re-tracing adjoint with 64-bit fixed-point
    accumulation instead of float atomics.  At the BASELINE size (512x512x64, material rows): two runs are bitwise equal — also under another
    workgroup-to-XCD mapping and another tile enumeration, i.e. whatever order the taps arrive in — and the float-atomic adjoints (re-traced,
    cached K9, forward + adjoint in one launch) agree with it within 1e-3 of the gradient's scale on all but 1e-3 of the texels; the same
    for the gaussian film.  At a small size it meets the oracle's (serial, double precision) sums to float rounding."""
    for k in ("FFX_XCD_REMAP", "FFX_TILE_BLOCK", "FFX_DETERMINISTIC"):
        monkeypatch.delenv(k, raising=False)
    from tests.test_bruteforce_cpu import material_rows

    sc = scenes.vocalfold()
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off)
    gd.update(_rand_xforms(2, 6), off + np.array([0, 9 * stride[1]], np.int32))
    mats = material_rows(2, 3, anisotropic=0.0)
    tex = _tex(sc)
    rng = np.random.default_rng(8)
    gimg = dev(np.where(rng.random((512, 512, 3)) < 0.5, -1.0, 1.0).astype(np.float32) / (512 * 512 * 3))
    for rf in ("box", "gaussian"):
        sd = scene_desc.scene_desc(sc, tex_channels=1, shadows=True, mat_stride=16, rfilter=rf)
        g_det = gd.render_bwd(sd, dev(mats), 64, 5, gimg, deterministic=True)
        assert torch.equal(g_det, gd.render_bwd(sd, dev(mats), 64, 5, gimg, deterministic=True))
        monkeypatch.setenv("FFX_XCD_REMAP", "1")  # one image band per XCD: a completely different arrival order
        monkeypatch.setenv("FFX_TILE_BLOCK", "2")
        assert torch.equal(g_det, gd.render_bwd(sd, dev(mats), 64, 5, gimg, deterministic=True)), rf
        monkeypatch.delenv("FFX_XCD_REMAP")
        monkeypatch.delenv("FFX_TILE_BLOCK")
        gs = float(g_det.abs().max())
        assert gs > 0 and bool(torch.isfinite(g_det).all())
        g_atomic = gd.render_bwd(sd, dev(mats), 64, 5, gimg, deterministic=False)
        cache = torch.zeros(ops.render_cache_bytes_sd(sd, 64), dtype=torch.uint8, device="cuda")
        gd.render_fwd(sd, dev(mats), tex, 64, seed=5, cache=cache)
        g_cached = gd.render_bwd_cached(sd, dev(mats), cache, 64, gimg, seed=5 if rf == "gaussian" else None)
        del cache
        _, g_fused = gd.render_fwd_adjoint(sd, dev(mats), tex, 64, 5, gimg)
        for what, g in (("re-traced", g_atomic), ("cached", g_cached), ("fused", g_fused)):
            err = (g - g_det).abs()
            assert float((err > 1e-3 * gs).float().mean()) <= 1e-3 and float(err.max()) <= 0.05 * gs, (rf, what, float(err.max()) / gs)
    # small: against the oracle, 1- and 3-channel textures, a Lambert table
    sc2 = scenes.vocalfold(width=52, height=44, tex=80, frames=3, n_fold=20, tube=(20, 24))
    go, gd2, alb2 = _pair(oracle, sc2, frame=2, xforms=_rand_xforms(2, 5))
    g2 = rng.standard_normal((44, 52, 3)).astype(np.float32)
    for ch in (1, 3):
        sd2 = scene_desc.scene_desc(sc2, tex_channels=ch, shadows=True)
        a = host(gd2.render_bwd(sd2, dev(alb2), 9, 3, dev(g2), deterministic=True))
        b = go.render_bwd(sd2, alb2, 9, 3, g2)
        sb = float(np.abs(b).max())
        assert sb > 0 and (np.abs(a - b) > 1e-4 * sb).mean() <= 1e-3 and np.abs(a - b).max() <= 0.1 * sb
    # nothing lit: gtex stays as it is
    sd0 = scene_desc.scene_desc(sc2, tex_channels=1, shadows=True)
    assert float(gd2.render_bwd(sd0, dev(alb2), 9, 3, torch.zeros(44, 52, 3, device="cuda"), deterministic=True).abs().max()) == 0.0


def test_per_slot_normal_area_holds_what_the_header_says():
    """include/ffx.h ffx_bvh_info.off_gn: per leaf slot the unit geometric normal and, in the fourth word, the BITS (shape + 1) | smooth << 30
    (0: degenerate triangle) — the render kernels take a hit's shape and smooth flag from that word.  The records of the same blob give the
    expectation (cross product of the two edges, normalised), for a scene with a smooth and a flat shape after a re-fit."""
    sc = scenes.vocalfold(width=32, height=32, tex=32, frames=3, n_fold=12, tube=(12, 16))
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(sc)
    gd = ops.DeviceGeometry(pool, tris, shape, off, smooth=[True] + [False] * (len(off) - 1))
    gd.update(_rand_xforms(len(off), 3), off)
    torch.cuda.synchronize()
    info, blob = gd.info, gd.blob
    F = int(info.n_tris)
    assert int(info.off_gn) > 0
    recs = blob[int(info.off_recs): int(info.off_recs) + 48 * F].cpu().numpy().view(np.float32).reshape(F, 12)
    gn = blob[int(info.off_gn): int(info.off_gn) + 16 * F].cpu().numpy()
    n_dev, word = gn.view(np.float32).reshape(F, 4)[:, :3], gn.view(np.uint32).reshape(F, 4)[:, 3]
    e1, e2 = recs[:, 3:6].astype(np.float64), recs[:, 6:9].astype(np.float64)
    shape_of_slot = recs[:, 10].view(np.int32)
    n = np.cross(e1, e2)
    ln = np.linalg.norm(n, axis=1)
    ok = ln > 0
    assert ok.mean() > 0.99
    np.testing.assert_allclose(n_dev[ok], (n[ok] / ln[ok, None]), rtol=0, atol=3e-6)
    assert np.array_equal(word[ok] & 0x3FFFFFFF, (shape_of_slot[ok] + 1).astype(np.uint32))
    assert np.array_equal((word[ok] >> 30) & 1, (shape_of_slot[ok] == 0).astype(np.uint32))  # shape 0 is the smooth one
    assert (word[~ok] == 0).all()
