"""Pins the CPU oracle (oracle/ffx_oracle.c) to golden vectors captured from the reference's
own torch code (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from tests.conftest import load_golden

FLIP_Y = np.diag([1.0, -1.0, 1.0, 1.0]).astype(np.float32)


def test_k1_projection_matches_reference(oracle):
    g = load_golden("g2_projection.npz")
    KF = g["K"] @ FLIP_Y
    for n in (8, 18):
        out = oracle.project_rays_fwd(g[f"rays_{n}"], KF)
        np.testing.assert_allclose(out, g[f"ndc_{n}"], rtol=2e-6, atol=2e-7)
        grays = oracle.project_rays_bwd(g[f"rays_{n}"], KF, g[f"gw_{n}"])
        np.testing.assert_allclose(grays, g[f"grays_{n}"], rtol=2e-5, atol=2e-6)
        # inverse (projectNDCPointsToWorld) = transform_points with inv(K@FLIP_Y)
        back = oracle.transform_points(g[f"ndc_{n}"], np.linalg.inv(KF.astype(np.float64)).astype(np.float32))
        np.testing.assert_allclose(back, g[f"back_{n}"], rtol=1e-3, atol=1e-4)


def test_transform_points_matches_reference(oracle):
    g = load_golden("g6_math.npz")
    np.testing.assert_allclose(oracle.transform_points(g["pts"], g["T"], 0), g["transform_points"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(oracle.transform_points(g["pts"], g["T"], 1), g["transform_directions"], rtol=2e-6, atol=1e-6)


CASES = ["a", "b", "c", "d", "e", "f", "g"]


@pytest.mark.parametrize("name", CASES)
def test_k2_dense_forward_and_backward(oracle, name):
    g = load_golden("g3_rasterize_points.npz")
    pts, (s0, s1), sigma = g[f"{name}_pts"], g[f"{name}_size"], float(g[f"{name}_sigma"])
    dense = oracle.splat_dense_fwd(pts, sigma, int(s0), int(s1))
    assert dense.shape == g[f"{name}_dense"].shape
    # fp32 expf implementations differ by <= 1-2 ulp; values are in [0,1]
    np.testing.assert_allclose(dense, g[f"{name}_dense"], rtol=0, atol=3e-7)
    gp = oracle.splat_dense_bwd(pts, sigma, int(s0), int(s1), g[f"{name}_dense_w"])
    ref = g[f"{name}_dense_gpts"]
    np.testing.assert_allclose(gp, ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["sum", "softor"])
def test_k2_fused_reduce_forward_and_backward(oracle, name, mode):
    g = load_golden("g3_rasterize_points.npz")
    pts, (s0, s1), sigma = g[f"{name}_pts"], g[f"{name}_size"], float(g[f"{name}_sigma"])
    red = 0 if mode == "sum" else 1
    tex = oracle.splat_fwd(pts, sigma, red, -1, int(s0), int(s1))
    ref = g[f"{name}_{mode}"]
    np.testing.assert_allclose(tex, ref, rtol=2e-6, atol=1e-6)
    gp = oracle.splat_bwd(pts, sigma, red, -1, int(s0), int(s1), tex, g[f"{name}_w"])
    refg = g[f"{name}_{mode}_gpts"]
    np.testing.assert_allclose(gp, refg, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(refg).max()))


def test_k2_survey_anchors(oracle):
    g = load_golden("g3_rasterize_points.npz")
    a = oracle.splat_dense_fwd(np.array([[0.25, 0.75]], np.float32), 4.0, 8, 16)
    assert a.shape == (1, 16, 8)
    assert np.unravel_index(a.argmax(), a.shape) == (0, 12, 2)
    np.testing.assert_allclose(a, g["anchor1"], atol=3e-7)
    b = oracle.splat_dense_fwd(g["anchor2_pts"], 10.0, 16, 16)
    np.testing.assert_allclose(b, g["anchor2"], atol=3e-7)
    assert abs(float(b.sum()) - 98.0697) < 1e-3


def test_k2_full_size_500(oracle):
    g = load_golden("g3_full_500.npz")
    for mode, red in (("sum", 0), ("softor", 1)):
        tex = oracle.splat_fwd(g["pts"], 10.0, red, -1, 500, 500)
        np.testing.assert_allclose(tex, g[mode], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["in", "bd"])
def test_k2_baked_variants(oracle, tag):
    g = load_golden("g4_baked.npz")
    pts = g[f"{tag}_pts"]
    # footprint = floor(sqrt(sigma)) * num_std, made odd (rasterization.py:180-182)
    half_sum = (10 * 4 + 1 - 1) // 2
    half_sor = ((10 * 5 + 1) - 1) // 2
    out = oracle.splat_fwd(pts, 100.0, 0, half_sum, 100, 100)
    np.testing.assert_allclose(out, g[f"{tag}_baked_sum"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out.T, g[f"{tag}_baked_sum_2"], rtol=2e-6, atol=1e-6)
    out = oracle.splat_fwd(pts, 100.0, 1, half_sor, 100, 100)
    np.testing.assert_allclose(out, g[f"{tag}_baked_softor"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out, g[f"{tag}_baked_softor_2"], rtol=2e-6, atol=1e-6)


def test_k2_baked_nonsquare_and_grad(oracle):
    g = load_golden("g4_baked.npz")
    s0, s1 = [int(v) for v in g["ns_size"]]
    half = (4 * 4 + 1 - 1) // 2
    out = oracle.splat_fwd(g["ns_pts"], 16.0, 0, half, s0, s1)
    assert out.shape == g["ns_baked_sum"].shape == (s1, s0)
    np.testing.assert_allclose(out, g["ns_baked_sum"], rtol=2e-6, atol=1e-6)
    out = oracle.splat_fwd(g["ns_pts"], 16.0, 1, (4 * 5 + 1 - 1) // 2, s0, s1)
    np.testing.assert_allclose(out, g["ns_baked_softor"], rtol=2e-6, atol=1e-6)
    tex = oracle.splat_fwd(g["in_pts"], 100.0, 0, 20, 100, 100)
    gp = oracle.splat_bwd(g["in_pts"], 100.0, 0, 20, 100, 100, tex, g["in_baked_sum_w"])
    ref = g["in_baked_sum_gpts"]
    # the weights sin(0.21 k) make this a heavily cancelling sum (|terms| ~ 1e-2, result ~ 3e-4):
    # a float64 evaluation of the same formula differs from the reference's own fp32 autograd
    # by 2e-5, so 4e-5 absolute is the pinning tolerance here.
    np.testing.assert_allclose(gp, ref, rtol=0, atol=4e-5)


def test_k2_depth_and_lines(oracle):
    g = load_golden("g5_depth_lines.npz")
    s0, s1 = [int(v) for v in g["depth_size"]]
    out = oracle.splat_depth_fwd(g["depth_pts"][:, :2], g["depth_pts"][:, 2], 6.0, s0, s1)
    np.testing.assert_allclose(out, g["depth_out"], rtol=2e-6, atol=1e-6)
    l0, l1 = [int(v) for v in g["lines_size"]]
    out = oracle.splat_lines_fwd(g["lines_in"], 3.0, l0, l1)
    np.testing.assert_allclose(out, g["lines_out"], rtol=1e-5, atol=1e-6)


def test_rasterize_lines_gradient_matches_the_reference_autograd(oracle):
    """rasterize_lines is differentiable in the reference (its line-regularisation loop optimises the segments
    through it, rasterization.py:645-743): forward and d/d(segments) against torch.autograd on the reference's own
    expression (oracle/gen_golden_r2.py), three sizes / sigmas."""
    g = load_golden("g10_lines_grad.npz")
    for t in "abc":
        s0, s1 = [int(v) for v in g[f"{t}_size"]]
        sigma = float(g[f"{t}_sigma"])
        out = oracle.splat_lines_fwd(g[f"{t}_lines"], sigma, s0, s1)
        np.testing.assert_allclose(out, g[f"{t}_out"], rtol=1e-5, atol=1e-6)
        gl = oracle.splat_lines_bwd(g[f"{t}_lines"], sigma, s0, s1, g[f"{t}_w"])
        ref = g[f"{t}_glines"]
        np.testing.assert_allclose(gl, ref, rtol=1e-4, atol=2e-6 * np.abs(ref).max())
    # a zero upstream gradient gives a zero result; no lines is a no-op
    z = oracle.splat_lines_bwd(g["a_lines"], 3.0, 20, 20, np.zeros_like(g["a_out"]))
    assert not z.any()
