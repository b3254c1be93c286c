"""Host-side logic of the reference-API mirror against golden vectors captured from the reference
(g1, g6, g7, g8) plus behaviour checks of Scene over a stand-in parameter object.  CPU only: none
of this touches a kernel."""
import random

import numpy as np
import pytest
import torch

import fireflies_amd as ff
from fireflies_amd import mi, scenes
from fireflies_amd.utils import math as M
from tests.conftest import load_golden

CPU = "cpu"


def test_g1_generate_uniform_rays():
    g = load_golden("g1_uniform_rays.npz")
    for n in (8, 16, 18, 32):
        r = ff.projection.Laser.generate_uniform_rays(0.0275, n, n, device=CPU).numpy()
        np.testing.assert_array_equal(r, g[f"rays_{n}"])
    r = ff.projection.Laser.generate_uniform_rays(0.0275, 18, 18, device=CPU)
    np.testing.assert_allclose(r[0].numpy(), [-0.2257, -0.2257, -0.9477], atol=1e-4)  # SURVEY App. C anchor
    # non-square grids get every row filled exactly once (the reference's index breaks here)
    r = ff.projection.Laser.generate_uniform_rays(0.05, 3, 5, device=CPU)
    assert r.shape == (15, 3) and torch.isfinite(r).all() and (r.norm(dim=1) - 1).abs().max() < 1e-6


def test_g6_math_helpers():
    g = load_golden("g6_math.npz")
    pts, T = torch.from_numpy(g["pts"]), torch.from_numpy(g["T"])
    np.testing.assert_array_equal(M.transform_points(pts, T).numpy(), g["transform_points"])
    np.testing.assert_array_equal(M.transform_directions(pts, T).numpy(), g["transform_directions"])
    np.testing.assert_array_equal(M.toMat4x4(T[:3, :3].clone()).numpy(), g["toMat4x4"])
    np.testing.assert_array_equal(M.toMat4x4(T[:3, :3].clone(), addOne=False).numpy(), g["toMat4x4_noone"])
    for nm in ("getYawTransform", "getPitchTransform", "getRollTransform", "getXTransform", "getYTransform", "getZTransform"):
        np.testing.assert_array_equal(getattr(M, nm)(0.4, CPU).numpy(), g[nm])
    torch.manual_seed(11)
    np.testing.assert_array_equal(M.randomBetweenTensors(torch.from_numpy(g["rbt_a"]), torch.from_numpy(g["rbt_b"])).numpy(), g["randomBetweenTensors"])
    np.testing.assert_array_equal(M.normalize(torch.from_numpy(g["normalize_in"])).numpy(), g["normalize"])
    R = M.rotation_matrix_from_vectors(torch.from_numpy(g["rmfv_v1"]), torch.from_numpy(g["rmfv_v2"]))
    np.testing.assert_allclose(R.numpy(), g["rotation_matrix_from_vectors"], atol=1e-6)
    # utils.transforms is empty in the reference; here it resolves
    from fireflies_amd.utils import transforms

    assert transforms.transform_points is M.transform_points


def test_g7_randomize_draw_order_and_axis_convention():
    g = load_golden("g7_randomize.npz")
    verts = torch.from_numpy(g["verts"])
    centroid = verts.sum(dim=0, keepdim=True) / verts.shape[0]
    E = ff.entity
    for s in (0, 1, 2):
        torch.manual_seed(s)
        t = E.Transformable("x", CPU)
        t.rotate_z(-1, 1)
        t.translate_x(-0.5, 0.5)
        t.train()
        t.randomize()
        np.testing.assert_array_equal(t.world().numpy(), g[f"tr_world_{s}"])

        torch.manual_seed(s)
        t = E.Transformable("y", CPU)
        t.rotate_x(-0.3, 0.3)
        t.rotate_y(-0.2, 0.4)
        t.rotate_z(0.1, 0.9)
        t.translate_x(-0.5, 0.5)
        t.translate_y(1.0, 2.0)
        t.translate_z(-3.0, -2.0)
        t.add_float_key("fkey", 1.0, 3.0)
        t.add_vec3_key("vkey", torch.tensor([0.0, 1.0, 2.0]), torch.tensor([1.0, 2.0, 3.0]))
        t.train()
        t.randomize()
        np.testing.assert_allclose(t.world().numpy(), g[f"tr_full_world_{s}"], rtol=0, atol=1e-7)
        np.testing.assert_array_equal(t.get_randomized_float_attributes()["fkey"].numpy(), g[f"tr_full_fkey_{s}"])
        np.testing.assert_array_equal(t.get_randomized_vec3_attributes()["vkey"].numpy(), g[f"tr_full_vkey_{s}"])

        torch.manual_seed(s)
        m = E.Mesh("mesh", verts - centroid, CPU)
        m.set_centroid(centroid)
        m.scale_x(0.5, 2.0)
        m.scale_z(1.0, 3.0)
        m.rotate_y(-0.25, 0.25)
        m.translate_y(-0.05, 0.05)
        m.train()
        m.randomize()
        np.testing.assert_allclose(m.world().numpy(), g[f"mesh_world_{s}"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(m.get_randomized_vertices().numpy(), g[f"mesh_verts_{s}"], rtol=1e-6, atol=1e-6)

        torch.manual_seed(s)
        parent = E.Mesh("parent", verts - centroid, CPU)
        parent.set_centroid(centroid)
        parent.rotate_x(-0.5, 0.5)
        child = E.Mesh("child", (verts - centroid) * 0.5, CPU)
        child.translate_z(0.1, 0.9)
        child.setParent(parent)
        parent.train()
        child.train()
        parent.randomize()
        child.randomize()
        np.testing.assert_allclose(parent.world().numpy(), g[f"pc_parent_world_{s}"], atol=1e-7)
        np.testing.assert_allclose(child.world().numpy(), g[f"pc_child_world_{s}"], atol=1e-7)
        np.testing.assert_allclose(child.get_randomized_vertices().numpy(), g[f"pc_child_verts_{s}"], rtol=1e-6, atol=1e-6)

        torch.manual_seed(s)
        am = E.Mesh("anim", verts - centroid, CPU)
        am.add_animation_func(lambda v, t_: v * (1.0 + t_), 0.0, 1.0)
        am.rotate_z(-0.2, 0.2)
        am.train()
        am.randomize()
        np.testing.assert_allclose(am.get_randomized_vertices().numpy(), g[f"anim_verts_{s}"], rtol=1e-6, atol=1e-6)
    # SURVEY App. C anchor: rotate_z rotates about the Y axis, translation is drawn first
    torch.manual_seed(0)
    t = E.Transformable("x", CPU)
    t.rotate_z(-1, 1)
    t.translate_x(-0.5, 0.5)
    t.randomize()
    w = t.world().numpy()
    assert abs(w[0, 3] - (-0.0037)) < 1e-4 and abs(w[0, 2] - 0.2650) < 1e-4 and w[1, 1] == 1.0


def test_g8_sampler_sequences():
    g = load_golden("g8_samplers.npz")
    S = ff.sampling
    smp = S.UniformSampler(0.0, 0.05, device=CPU)
    smp.eval()
    np.testing.assert_array_equal(np.array([float(smp.sample()) for _ in range(12)], np.float32), g["uniform_scalar_eval"])
    np.testing.assert_array_equal(smp.get_min().numpy(), g["uniform_scalar_min_after"])  # the documented drift
    smp = S.UniformSampler(torch.tensor([0.0, 1.0, 2.0]), torch.tensor([0.03, 1.03, 2.03]), device=CPU)
    smp.eval()
    np.testing.assert_array_equal(np.stack([smp.sample().clone().numpy() for _ in range(8)]), g["uniform_vec3_eval"])
    smp = S.UniformSampler(torch.tensor([0.0, 1.0, 2.0]), torch.tensor([0.03, 1.0, 2.0]), device=CPU)
    smp.eval()
    np.testing.assert_array_equal(np.stack([smp.sample().clone().numpy() for _ in range(8)]), g["uniform_vec3_degenerate_eval"])
    smp = S.UniformSampler(torch.ones(3), torch.ones(3), device=CPU)
    smp.eval()
    np.testing.assert_array_equal(np.stack([smp.sample().clone().numpy() for _ in range(3)]), g["uniform_const_eval"])
    an = S.AnimationSampler(0, 1, 0, 1, device=CPU)
    an.set_eval_interval(0, 5)
    an.eval()
    np.testing.assert_array_equal(np.array([an.sample() for _ in range(14)]), g["animation_eval"])
    an = S.AnimationSampler(0, 7, 0, 3, device=CPU)
    an.train()
    random.seed(42)
    np.testing.assert_array_equal(np.array([an.sample() for _ in range(20)]), g["animation_train_seed42"])
    torch.manual_seed(21)
    sv = S.UniformScalarToVec3Sampler(1.0, 20.0, device=CPU)
    sv.train()
    np.testing.assert_array_equal(np.stack([sv.sample().numpy() for _ in range(4)]), g["scalar_to_vec3_train_seed21"])
    sv.eval()
    np.testing.assert_array_equal(np.stack([sv.sample().numpy() for _ in range(4)]), g["scalar_to_vec3_eval"])
    torch.manual_seed(22)
    gs = S.GaussianSampler(torch.tensor([0.0]), torch.tensor([1.0]), torch.tensor([0.5, 0.5]), torch.tensor([0.1, 0.2]), device=CPU)
    gs.train()
    np.testing.assert_array_equal(np.stack([gs.sample().numpy() for _ in range(4)]), g["gaussian_train_seed22"])
    # the integer sampler constructs (it raises in the reference) and sweeps / draws in range
    ui = S.UniformIntegerSampler(2, 6, device=CPU)
    ui.eval()
    assert [ui.sample() for _ in range(6)] == [2, 3, 4, 5, 2, 3]
    ui.train()
    assert all(2 <= ui.sample() < 6 for _ in range(50))


class FakeParams(dict):
    """stand-in for mi.SceneParameters WITHOUT the device fast path: exercises the reference code
    path of Scene (vertex tensors assigned to `<mesh>.vertex_positions`)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.updates = 0

    def update(self):  # noqa: A003
        self.updates += 1


def _fake_scene_params():
    cube_v, _ = scenes.make_cube((1.0, 2.0, 3.0), 0.5)
    p = FakeParams()
    p["mesh-Cube.vertex_positions"] = mi.Float32(torch.from_numpy(cube_v.reshape(-1)))
    p["PerspectiveCamera.to_world"] = mi.Transform4f(scenes.look_at((0, 0, 0), (0, 0, 1)))
    p["PerspectiveCamera.x_fov"] = mi.Float(40.0)
    p["PerspectiveCamera_1.to_world"] = mi.Transform4f(scenes.look_at((1, 0, 0), (0, 0, 1)))
    p["Projector.to_world"] = mi.Transform4f(scenes.look_at((1, 0, 0), (0, 0, 1)))
    p["emit-Spot.to_world"] = mi.Transform4f(np.eye(4, dtype=np.float32))
    p["emit-Spot.intensity.value"] = mi.Color3f([5.0, 5.0, 5.0])
    p["emit-Spot.cutoff_angle"] = mi.Float(20.0)
    p["mat-Default.brdf_0.base_color.value"] = mi.Color3f([0.8, 0.2, 0.3])
    p["mat-Default.brdf_0.specular"] = mi.Float(0.5)
    p["tex.data"] = mi.TensorXf(torch.zeros(4, 4, 3))
    return p, cube_v


def test_scene_classification_and_randomize_over_generic_params():
    p, cube_v = _fake_scene_params()
    sc = ff.Scene(p, device=CPU)
    assert [m.name() for m in sc.meshes()] == ["mesh-Cube"]
    assert sc._camera.name() == "PerspectiveCamera_1"  # last camera-like key in sorted order wins, like the reference
    assert sc._projector.name() == "Projector"
    assert [l.name() for l in sc.lights()] == ["emit-Spot"] and [m.name() for m in sc.materials()] == ["mat-Default"]
    assert set(sc.light("emit-Spot").float_attributes()) == {"cutoff_angle"}
    assert set(sc.light("emit-Spot").vec3_attributes()) == {"intensity.value"}
    assert sc.mesh("nope") is None
    # nothing is randomisable yet: randomize() only calls update()
    sc.randomize()
    assert p.updates == 1
    mesh = sc.mesh("mesh-Cube")
    mesh.scale_x(0.5, 2.0)
    mesh.rotate_y(-0.25, 0.25)
    light = sc.light("emit-Spot")
    light.add_vec3_sampler("intensity.value", ff.sampling.UniformScalarToVec3Sampler(1.0, 20.0, device=CPU))
    mat = sc.material("mat-Default")
    mat.add_float_key("brdf_0.specular", 0.0, 0.75)
    sc._camera.translate_x(-0.1, 0.1)
    sc.train()
    torch.manual_seed(3)
    sc.randomize()
    # replay the same stream by hand.  Every sampler draws even when its range is degenerate:
    # mesh (t, r, s); light (t, r, float attrs, vec3 attrs); material (float attrs, vec3 attrs);
    # camera (t, r)
    torch.manual_seed(3)
    t = torch.rand(3) * 0.0
    r = torch.rand(3) * torch.tensor([0.0, 0.5, 0.0]) + torch.tensor([0.0, -0.25, 0.0])
    s = torch.rand(3) * torch.tensor([1.5, 0.0, 0.0]) + torch.tensor([0.5, 1.0, 1.0])
    torch.rand(3), torch.rand(3), torch.rand(1)  # light: translation, rotation, cutoff_angle
    inten = float(torch.rand(1) * 19.0 + 1.0)
    spec = float(torch.rand(1) * 0.75)
    torch.rand(3)  # material base_color (degenerate range)
    cam_t = torch.rand(3) * torch.tensor([0.2, 0.0, 0.0]) + torch.tensor([-0.1, 0.0, 0.0])
    c = cube_v.mean(0)
    Ry = M.getYawTransform(float(r[1]), CPU)  # rotate_y -> "Yaw" = about Z (the reference's convention)
    W = np.eye(4, dtype=np.float32)
    W[:3, :3] = (Ry @ torch.diag(s)).numpy()
    W[:3, 3] = c
    expect = (cube_v - c) @ W[:3, :3].T + W[:3, 3]
    got = p["mesh-Cube.vertex_positions"].torch().reshape(-1, 3).numpy()
    np.testing.assert_allclose(got, expect, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(p["emit-Spot.intensity.value"].torch().reshape(-1).numpy(), [inten] * 3, rtol=1e-6)
    assert abs(float(p["mat-Default.brdf_0.specular"]) - spec) < 1e-6
    cw = p["PerspectiveCamera_1.to_world"].matrix.torch()[0].numpy()
    base = scenes.look_at((1, 0, 0), (0, 0, 1))
    np.testing.assert_allclose(cw[:3, 3], base[:3, 3] + cam_t.numpy(), atol=1e-6)
    assert p.updates == 2
    # eval mode is deterministic
    sc.eval()
    sc.randomize()
    a = p["mesh-Cube.vertex_positions"].torch().clone()
    sc2 = ff.Scene(_fake_scene_params()[0], device=CPU)
    sc2.mesh("mesh-Cube").scale_x(0.5, 2.0)
    sc2.mesh("mesh-Cube").rotate_y(-0.25, 0.25)
    sc2.eval()
    sc2.randomize()
    torch.testing.assert_close(a, sc2._mitsuba_params["mesh-Cube.vertex_positions"].torch())


def _randomisable_fake_scene():
    p, _ = _fake_scene_params()
    sc = ff.Scene(p, device=CPU)
    mesh = sc.mesh("mesh-Cube")
    mesh.scale_x(0.5, 2.0)
    mesh.rotate_y(-0.25, 0.25)
    mesh.translate_z(-0.3, 0.3)
    sc.light("emit-Spot").add_vec3_sampler("intensity.value", ff.sampling.UniformScalarToVec3Sampler(1.0, 20.0, device=CPU))
    sc.material("mat-Default").add_float_key("brdf_0.specular", 0.0, 0.75)
    sc._camera.translate_x(-0.1, 0.1)
    sc.train()
    return p, sc


def _snapshot(p):
    return {k: (v.torch().clone() if hasattr(v, "torch") and not isinstance(v, mi.Float) else (v.matrix.torch().clone() if hasattr(v, "matrix") else float(v)))
            for k, v in p.items() if k != "tex.data"}


def test_randomize_is_one_transfer_and_batches_replay_the_sequential_draws(monkeypatch):
    """f1: Scene.randomize() draws everything first (reference order) and fetches it with ONE device-to-host
    transfer (it was one .tolist() per draw); randomize_batch(seeds) draws S scene samples up front under their
    own seeds — same numbers as S sequential `manual_seed(s); random.seed(s); randomize()` calls — again with one
    transfer, and applies them one by one."""
    fetches = []
    orig = ff.entity.DrawBatch.start_fetch
    monkeypatch.setattr(ff.entity.DrawBatch, "start_fetch", lambda self: (fetches.append(len(self)), orig(self))[1])
    seeds = [11, 12, 13, 14]
    p1, sc1 = _randomisable_fake_scene()
    seq = []
    for s_ in seeds:
        torch.manual_seed(s_)
        random.seed(s_)
        sc1.randomize()
        seq.append(_snapshot(p1))
    assert len(fetches) == len(seeds) and all(n >= 10 for n in fetches)  # mesh t,r,s + light t,r,2 attrs + material 2 attrs + camera t,r
    fetches.clear()
    p2, sc2 = _randomisable_fake_scene()
    appliers = sc2.randomize_batch(seeds)
    assert len(fetches) == 1 and len(appliers) == len(seeds)
    for apply_k, ref in zip(appliers, seq):
        apply_k()
        got = _snapshot(p2)
        assert got.keys() == ref.keys()
        for k in ref:
            if isinstance(ref[k], float):
                assert got[k] == ref[k], k
            else:
                torch.testing.assert_close(got[k], ref[k], rtol=0, atol=0, msg=k)
    assert p2.updates == len(seeds)
    # the poses of different seeds really differ
    assert not torch.equal(seq[0]["mesh-Cube.vertex_positions"], seq[1]["mesh-Cube.vertex_positions"])


def test_sampler_host_mirrors_follow_in_place_writes_through_an_old_handle():
    """a draw through entity.DrawBatch uses float32 host mirrors of the sampler's bounds; a caller that keeps the tensor get_min() handed
    out and writes to it LATER (`lo += 1`) changes the bounds the reference's sampler reads live — the mirrors follow (tensor version
    counters), so batched draws and plain sample() keep agreeing"""
    from fireflies_amd import entity
    from fireflies_amd import sampling as S

    smp = S.UniformSampler(torch.tensor([0.0, 10.0]), torch.tensor([1.0, 11.0]), device="cpu")
    lo_handle, hi_handle = smp.get_min(), smp.get_max()

    def batched():
        torch.manual_seed(3)
        b = entity.DrawBatch()
        i = smp.draw(b)
        return np.asarray(b.fetch()[i], np.float32).reshape(-1)

    def plain():
        torch.manual_seed(3)
        return smp.sample().numpy().reshape(-1)

    np.testing.assert_array_equal(batched(), plain())
    lo_handle += 5.0  # (no accessor is called: only the tensor's version moves)
    hi_handle.mul_(2.0).add_(10.0)
    after = batched()
    np.testing.assert_array_equal(after, plain())
    assert (after >= np.float32([5.0, 15.0])).all() and (after <= np.float32([12.0, 32.0])).all()


def test_material_has_no_pose_but_warns():
    m = ff.material.Material("mat-X", device=CPU)
    with pytest.warns(UserWarning):
        m.rotate_x(0.0, 1.0)
    m.add_float_key("a", 1.0, 2.0)
    m.randomize()
    assert 1.0 <= float(m.get_randomized_float_attributes()["a"]) <= 2.0


def test_mi_conventions():
    K = mi.perspective_projection((500, 500), (500, 500), (0, 0), 30.0, 0.01, 100.0).numpy()
    np.testing.assert_allclose(K, scenes.perspective_projection(500, 500, 30.0, 0.01, 100.0), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(K, load_golden("g2_projection.npz")["K"], rtol=1e-6, atol=1e-7)
    K2 = mi.perspective_projection((640, 480), (640, 480), (0, 0), 60.0, 0.1, 10.0).numpy()
    np.testing.assert_allclose(K2, scenes.perspective_projection(640, 480, 60.0, 0.1, 10.0), rtol=1e-6, atol=1e-7)
    # a point on the optical axis lands in the middle of sample space; +x_cam maps to smaller u
    q = K2 @ np.array([0, 0, 2.0, 1.0])
    np.testing.assert_allclose(q[:2] / q[3], [0.5, 0.5], atol=1e-6)
    q = K2 @ np.array([0.5, 0, 2.0, 1.0])
    assert q[0] / q[3] < 0.5
    tw = scenes.look_at((1, 2, 3), (1, 2, 4))
    np.testing.assert_allclose(tw[:3, 2], [0, 0, 1], atol=1e-7)
    t = mi.Transform4f(tw.tolist())
    assert t.matrix.torch().shape == (1, 4, 4)
    assert isinstance(type(mi.Float(2.0))(3.5), mi.Float) and len(mi.Color3f([1, 2, 3])) == 3


def test_bridson_matches_the_reference_draw_for_draw():
    """sampling/poisson.py:16-116 draws from numpy's GLOBAL generator; under the same np.random.seed the
    samples are the reference's (tests/golden/g11_bridson.npz, oracle/gen_golden_r2.py): constant and
    spatially varying radius, k, the normal-distributed radius type, and generate_blue_noise_rays on top."""
    from fireflies_amd.sampling import poisson

    g = load_golden("g11_bridson.npz")
    for tag in ("const", "small"):
        np.random.seed(int(g[f"{tag}_seed"]))
        n, pts = poisson.bridson(np.ones(tuple(g[f"{tag}_shape"])) * float(g[f"{tag}_radius"]))
        assert n == int(g[f"{tag}_n"])
        np.testing.assert_array_equal(pts, g[f"{tag}_pts"])
    np.random.seed(13)
    n, pts = poisson.bridson(g["vary_map"], k=12)
    assert n == int(g["vary_n"])
    np.testing.assert_array_equal(pts, g["vary_pts"])
    np.random.seed(14)
    n, pts = poisson.bridson(np.ones((30, 30)) * 4.0, k=20, radiusType="normDist")
    assert n == int(g["norm_n"])
    np.testing.assert_array_equal(pts, g["norm_pts"])
    np.random.seed(21)
    rays = ff.projection.Laser.generate_blue_noise_rays(32, 24, 16, torch.from_numpy(g["bn_K"]), device=CPU)
    np.testing.assert_allclose(rays.numpy(), g["bn_rays"], rtol=1e-6, atol=2e-7)
    # the defining property of the reference's acceptance rule: no two samples within the square of
    # half-width ceil(r) cells (so their Euclidean distance exceeds ceil(r) - 1)
    rng = np.random.default_rng(0)
    n, pts = poisson.bridson(np.ones((60, 40)) * 6.0, rng=rng)
    assert n == len(pts) > 20
    cells = np.floor(pts).astype(int)
    cheb = np.abs(cells[:, None] - cells[None]).max(-1) + np.eye(n, dtype=int) * 10**6
    assert cheb.min() > 6
    assert (pts[:, 0] >= 0).all() and (pts[:, 0] <= 60).all() and (pts[:, 1] <= 40).all()
    rays = ff.projection.Laser.generate_blue_noise_rays(100, 100, 16, torch.from_numpy(scenes.perspective_projection(100, 100, 30.0, 0.01, 100.0)), device=CPU)
    assert rays.shape[1] == 3 and (rays[:, 2] < 0).all() and (rays.norm(dim=1) - 1).abs().max() < 1e-5


def test_laser_world_space_accessors_have_the_intended_values():
    """Laser.rays / origin / originPerRay raise AttributeError at the reference's HEAD (laser.py:163-177 read a
    non-existent attribute); the intended semantics — directions through transform_directions(_rays, world),
    origin = the projector's world matrix / its translation — are pinned here against utils.math (golden g6)."""
    tr = ff.entity.Transformable("projector", CPU)
    world = torch.tensor([[0.0, 0.0, 1.0, 0.3], [0.0, 1.0, 0.0, -0.2], [-1.0, 0.0, 0.0, 1.5], [0.0, 0.0, 0.0, 1.0]])
    tr.set_world(world)
    K = torch.from_numpy(scenes.perspective_projection(500, 500, 30.0, 0.01, 100.0))
    rays = ff.projection.Laser.generate_uniform_rays(0.05, 3, 3, device=CPU)
    laser = ff.projection.Laser(tr, rays, K, 30.0, 0.01, 100.0, device=CPU)
    got = laser.rays()
    np.testing.assert_allclose(got.numpy(), (rays @ world[:3, :3].T).numpy(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(got, ff.utils.math.transform_directions(rays, world))
    np.testing.assert_allclose(laser.origin().numpy(), world.numpy())
    op = laser.originPerRay()
    assert op.shape == rays.shape
    np.testing.assert_allclose(op.numpy(), np.tile(world[:3, 3].numpy(), (9, 1)))
    # the central ray (0, 0, -1) of the local frame maps to -(third column of the rotation)
    np.testing.assert_allclose(got[4].numpy(), -world[:3, 2].numpy(), atol=1e-6)
    laser.setToWorld(torch.eye(4))
    np.testing.assert_allclose(laser.rays().numpy(), rays.numpy(), atol=1e-7)


def test_laser_static_generators_and_yaml(tmp_path):
    K = torch.from_numpy(scenes.perspective_projection(500, 500, 30.0, 0.01, 100.0))
    r = ff.projection.Laser.generate_uniform_rays_by_count(4, 4, K, device=CPU)
    assert r.shape == (16, 3) and (r[:, 2] < 0).all()
    torch.manual_seed(0)
    r = ff.projection.Laser.generate_random_rays(10, K, device=CPU)
    assert r.shape == (10, 3) and (r.norm(dim=1) - 1).abs().max() < 1e-5
    # a ray through the screen centre is the optical axis
    c = ff.projection.Laser._unproject(torch.tensor([[0.5, 0.5, -1.0]]), K)
    np.testing.assert_allclose(c.numpy(), [[0, 0, -1]], atol=1e-6)


def test_intersection_helpers_match_reference():
    """fireflies/utils/intersections.py rayPlane / sphereSphere against golden g12 (captured from the reference,
    oracle/gen_golden_r2b.py) — including the exactly parallel ray, where the reference's `denom / denom` is NaN."""
    from fireflies_amd.utils import intersections

    g = load_golden("g12_dataset_helpers.npz")
    t = intersections.rayPlane(torch.from_numpy(g["rp_o"]), torch.from_numpy(g["rp_d"]), torch.from_numpy(g["rp_po"]), torch.from_numpy(g["rp_pn"]))
    np.testing.assert_array_equal(t.numpy(), g["rp_t"])
    assert np.isnan(g["rp_t"][3, 0]) and np.isfinite(np.delete(g["rp_t"], 3, axis=0)).all()
    hit = intersections.sphereSphere(torch.from_numpy(g["ss_a"]), torch.from_numpy(g["ss_ar"]), torch.from_numpy(g["ss_b"]), torch.from_numpy(g["ss_br"]))
    np.testing.assert_array_equal(hit.numpy(), g["ss_hit"])
    assert 0 < int(g["ss_hit"].sum()) < g["ss_hit"].size


def test_noise_texture_sampler_matches_the_reference():
    """sampling.NoiseTextureLerpSampler (fireflies/sampling/noise_texture_lerp.py): the base-colour textures of the reference's
    dataset loop (main.py:138-153), under the same torch / random seeds — golden g13 (every 8th texel + mean + sum of squares
    of two consecutive draws; a 512-wide texture is the smallest the 64 * 2^3 lattice admits)."""
    import random

    from fireflies_amd.sampling import NoiseTextureLerpSampler

    g = load_golden("g13_noise_texture.npz")
    for tag in "abc":
        seed, shape = int(g[f"{tag}_seed"]), tuple(int(v) for v in g[f"{tag}_shape"])
        torch.manual_seed(seed)
        random.seed(seed)
        ca, cb = torch.rand(3), torch.rand(3)
        np.testing.assert_array_equal(ca.numpy(), g[f"{tag}_color_a"])
        smp = NoiseTextureLerpSampler(color_a=ca, color_b=cb, texture_shape=shape, device=torch.device("cpu"))
        for k in range(2):
            t = smp.sample().numpy()
            assert t.shape == (3,) + shape
            np.testing.assert_allclose(t[:, ::8, ::8], g[f"{tag}_tex{k}_sub"], rtol=0, atol=1e-6)
            assert float(t.astype(np.float64).mean()) == pytest.approx(float(g[f"{tag}_tex{k}_mean"]), abs=1e-7)
            assert float((t.astype(np.float64) ** 2).sum()) == pytest.approx(float(g[f"{tag}_tex{k}_sq"]), rel=1e-6)
    from fireflies_amd.sampling.noise_texture_lerp import perlin_2d

    with pytest.raises(ValueError):  # fewer texels than lattice cells (the reference fails with a shape error deep inside)
        perlin_2d((16, 16), (32, 32))


def test_native_scene_randomiser_is_the_python_mirror_bit_for_bit(oracle):
    """f1 (include/ffx.h ffx_scene_randomize_h): the draws, interval maps, 4x4 algebra and parent chains of a whole randomisation in one
    host call.  Product library and oracle restatement against the Python mirror's own arithmetic (entity.Transformable / Mesh: numpy
    products through ffx_mat4_mul_h, math.cos / sin) on random configurations — bit for bit, which is what lets Scene.randomize() switch
    between the native call and the Python path without changing a seeded run, on any host CPU."""
    import ctypes as C

    from fireflies_amd import _abi, _lib
    from fireflies_amd.entity import Mesh, Transformable

    rng = np.random.default_rng(5)

    def fma_mm(A, B):
        """the definition (include/ffx.h ffx_mat4_mul_h): every element an fma chain over k, the first product plain — evaluated here in
        float64 with one rounding per step (a float32 product is exact in float64), independent of any BLAS"""
        A, B = np.asarray(A, np.float32), np.asarray(B, np.float32)
        out = np.zeros((4, 4), np.float32)
        for i in range(4):
            for j in range(4):
                acc = np.float32(A[i, 0] * B[0, j])
                for k in range(1, 4):
                    acc = np.float32(np.float64(A[i, k]) * np.float64(B[k, j]) + np.float64(acc))
                out[i, j] = acc
        return out

    from fireflies_amd.entity.base import mm4

    for _ in range(50):  # the mirror's product routine IS that definition (numpy's own 4x4 product rounds differently from CPU to CPU)
        A, B = rng.standard_normal((4, 4)).astype(np.float32), rng.standard_normal((4, 4)).astype(np.float32)
        np.testing.assert_array_equal(mm4(A, B), fma_mm(A, B))
        np.testing.assert_array_equal(mm4(torch.from_numpy(A), B), fma_mm(A, B))
    libs = {"hip": _lib.api().lib, "oracle": oracle.api().lib}
    for trial in range(20):
        n_ents = int(rng.integers(1, 6))
        draws, ents, spec = [], [], []
        for e in range(n_ents):
            kind = int(rng.integers(0, 3))
            r = _abi.RandEntity()
            r.kind, r.parent, r.draw_t, r.draw_r, r.draw_s = kind, (int(rng.integers(-1, e)) if e > 0 else -1), -1, -1, -1
            world = rng.standard_normal((4, 4)).astype(np.float32)
            world[3] = (0, 0, 0, 1)
            cen = rng.standard_normal(3).astype(np.float32)
            for j in range(16):
                r.world[j] = float(world.reshape(-1)[j])
            for j in range(3):
                r.centroid[j] = float(cen[j])
            randomised = kind != 0 and rng.random() < 0.8

            def add(n):
                d = _abi.RandDraw()
                d.n = n
                lo = rng.uniform(-2, 1, n).astype(np.float32)
                hi = (lo + rng.uniform(0, 2, n)).astype(np.float32)
                for j in range(n):
                    d.lo[j], d.hi[j] = float(lo[j]), float(hi[j])
                draws.append((d, lo, hi))
                return len(draws) - 1

            if randomised:
                r.draw_t, r.draw_r = add(3), add(3)
                if kind == 2:
                    r.draw_s = add(3)
            for _k in range(int(rng.integers(0, 3))):
                add(int(rng.integers(1, 5)))  # attribute draws in between
            ents.append(r)
            spec.append((kind, r.parent, r.draw_t, r.draw_r, r.draw_s, world, cen))
        nd, S = len(draws), 3
        seeds = (C.c_uint64 * S)(*[int(v) for v in rng.integers(0, 2**40, S)])
        offs = (C.c_uint64 * S)(*[4 * int(v) for v in rng.integers(0, 1000, S)])
        darr = (_abi.RandDraw * max(nd, 1))(*[d for d, _, _ in draws])
        earr = (_abi.RandEntity * n_ents)(*ents)
        outs = {}
        for name, lib in libs.items():
            vals = np.full((S, max(nd, 1), 4), np.nan, np.float32)
            mats = np.full((3, S, n_ents, 16), np.nan, np.float32)
            fn = lib.ffx_scene_randomize_h
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(_abi.RandDraw), C.c_int, C.POINTER(_abi.RandEntity), C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            assert fn(S, seeds, offs, darr, nd, earr, n_ents, vals.ctypes.data, mats[0].ctypes.data, mats[1].ctypes.data, mats[2].ctypes.data) == 0
            outs[name] = (vals, mats)
        np.testing.assert_array_equal(outs["hip"][0], outs["oracle"][0])
        np.testing.assert_array_equal(outs["hip"][1], outs["oracle"][1])
        vals, mats = outs["hip"]
        rand = libs["hip"].ffx_torch_rand_h
        rand.restype = C.c_int
        rand.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        for s in range(S):
            want_vals = []
            for d, (dr, lo, hi) in enumerate(draws):
                buf, inc = (C.c_float * 4)(), C.c_uint64()
                assert rand(seeds[s], offs[s] + 4 * d, dr.n, buf, C.byref(inc)) == 0
                u = np.asarray(buf[: dr.n], np.float32)
                v = u * (hi - lo) + lo  # (DrawBatch._values: float32, one rounding per operation)
                want_vals.append(v)
                np.testing.assert_array_equal(vals[s, d, : dr.n], v)
            chain = []
            for e, (kind, parent, dt, dr_, ds, world, cen) in enumerate(spec):
                if kind == 0 or dt < 0:
                    local = world
                else:
                    t = Transformable._translation_matrix(*[float(x) for x in want_vals[dt]]).numpy()
                    cm = np.zeros((4, 4), np.float32)
                    cm[:3, 3] = cen
                    rot = Transformable._rotation_matrix(*[float(x) for x in want_vals[dr_]]).numpy()
                    if kind == 2:
                        sc = np.zeros((4, 4), np.float32)
                        sc[0, 0], sc[1, 1], sc[2, 2], sc[3, 3] = *[float(x) for x in want_vals[ds]], 1.0
                        local = fma_mm(fma_mm(fma_mm(t + cm, rot), sc), world)  # entity/mesh.py Mesh._compose
                    else:
                        local = fma_mm(fma_mm(t + cm, rot), world)  # entity/base.py Transformable._compose
                w = local if parent < 0 else fma_mm(chain[parent], local)
                chain.append(w)
                unc = np.eye(4, dtype=np.float32)
                unc[:3, 3] = -cen
                np.testing.assert_array_equal(mats[0, s, e].reshape(4, 4), local)
                np.testing.assert_array_equal(mats[1, s, e].reshape(4, 4), w)
                np.testing.assert_array_equal(mats[2, s, e].reshape(4, 4), fma_mm(w, unc))  # Scene.update_meshes
    assert Mesh is not None


def test_native_params_update_writes_what_the_key_writes_write(oracle):
    """include/ffx.h ffx_scene_step_h (ABI 8) without a device (geom NULL: description and tables only): product library and oracle against the
    key writes spelled out in numpy — to_world blocks from the chain rows, spot attributes, material cells with Mitsuba's specular -> eta in
    double, mesh transforms from the un-centred chain, frame offsets — on random op tables; later ops overwrite earlier ones; a refused
    call (op out of range, frame out of range, material table missing) leaves every table as it was."""
    import ctypes as C

    from fireflies_amd import _abi, _lib

    rng = np.random.default_rng(11)
    words = C.sizeof(_abi.SceneDesc) // 4
    libs = {"hip": _lib.api().lib, "oracle": oracle.api().lib}
    for name, lib in libs.items():
        fn = lib.ffx_scene_step_h
        fn.restype, fn.argtypes = _abi.PROTOTYPES["ffx_scene_step_h"]

    def run(lib, plan, vals, chain, unc, frames, tmpl, mat, xf, off):
        sd = _abi.SceneDesc()
        mat, xf, off = mat.copy(), xf.copy(), off.copy()
        rc = lib.ffx_scene_step_h(plan, vals.ctypes.data, chain.ctypes.data, unc.ctypes.data, frames, tmpl, sd, mat.ctypes.data if mat.size else None,
                                  xf.ctypes.data, off.ctypes.data, None, 1, None)
        return rc, np.frombuffer(C.string_at(C.addressof(sd), C.sizeof(sd)), np.uint8).copy(), mat, xf, off

    for trial in range(30):
        S, nd, ne = int(rng.integers(1, 6)), int(rng.integers(1, 9)), int(rng.integers(1, 7))
        stride = 16 if trial % 2 else 3
        n_mat = S * stride
        vals = rng.uniform(0.01, 0.9, (nd, 4)).astype(np.float32)
        chain, unc = rng.standard_normal((ne, 16)).astype(np.float32), rng.standard_normal((ne, 16)).astype(np.float32)
        tmpl = _abi.SceneDesc()
        C.memmove(C.addressof(tmpl), rng.integers(0, 255, C.sizeof(tmpl), dtype=np.uint8).tobytes(), C.sizeof(tmpl))
        tmpl.n_mat_h = n_mat if trial % 3 else 0
        mat = rng.uniform(0, 1, n_mat).astype(np.float32)
        xf, off = rng.standard_normal((S, 16)).astype(np.float32), rng.integers(0, 1000, S).astype(np.int32)
        base, fstride, nfr = (rng.integers(0, 1000, S).astype(np.int32), rng.integers(1, 50, S).astype(np.int32), rng.integers(1, 9, S).astype(np.int32))
        ops = []
        for _ in range(int(rng.integers(0, 14))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                ops.append((0, int(rng.integers(0, ne)), 0, int(rng.integers(0, words - 16 + 1)), 0, 0))
            elif kind == 1:
                ops.append((1, int(rng.integers(0, nd)), int(rng.integers(0, 4)), int(rng.integers(0, words)), 0, 0))
            elif kind == 2:
                ops.append((2, int(rng.integers(0, nd)), int(rng.integers(0, 4)), int(rng.integers(0, n_mat)), int(rng.integers(0, 2)), 0))
            else:
                ops.append((3, int(rng.integers(0, ne)), 0, int(rng.integers(0, S)), 0, int(rng.integers(0, 2))))
        oarr = (_abi.StepOp * max(len(ops), 1))()
        for o, t in zip(oarr, ops):
            o.kind, o.src, o.comp, o.dst, o.conv, o.mode = t
        plan = _abi.StepPlan()
        plan.ops, plan.n_ops, plan.n_shapes, plan.n_draws, plan.n_ents = oarr, len(ops), S, nd, ne
        plan.frame_base, plan.frame_stride, plan.n_frames = (a.ctypes.data_as(C.POINTER(C.c_int32)) for a in (base, fstride, nfr))
        plan.n_mat_floats = n_mat
        fr = np.array([int(rng.integers(-1, nfr[s_])) for s_ in range(S)], np.int32)
        frames = (C.c_int32 * S)(*[int(v) for v in fr])
        # the key writes, spelled out
        w = np.frombuffer(C.string_at(C.addressof(tmpl), C.sizeof(tmpl)), np.uint8).copy().view(np.float32)
        want_mat, want_xf, want_off = mat.copy(), xf.copy(), off.copy()
        for kind, src, comp, dst, conv, mode in ops:
            if kind == 0:
                w[dst:dst + 16] = chain[src]
            elif kind == 1:
                w[dst] = vals[src, comp]
            elif kind == 2:
                v = float(vals[src, comp])
                want_mat[dst] = np.float32(2.0 / (1.0 - float(np.sqrt(0.08 * v))) - 1.0) if conv else vals[src, comp]
            else:
                want_xf[dst] = (chain if mode else unc)[src]
        want_sd = w.view(np.uint8).copy()
        if tmpl.n_mat_h > 0:
            o0 = _abi.SceneDesc.mat_h.offset
            want_sd[o0:o0 + 4 * n_mat] = want_mat.view(np.uint8)
        for s_ in range(S):
            if fr[s_] >= 0:
                want_off[s_] = base[s_] + fr[s_] * fstride[s_]
        for name, lib in libs.items():
            rc, sd, m2, x2, o2 = run(lib, plan, vals, chain, unc, frames, tmpl, mat, xf, off)
            assert rc == 0, (name, trial)
            np.testing.assert_array_equal(sd, want_sd, err_msg=name)
            np.testing.assert_array_equal(m2.view(np.uint32), want_mat.view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(x2.view(np.uint32), want_xf.view(np.uint32), err_msg=name)
            np.testing.assert_array_equal(o2, want_off, err_msg=name)
        # refusals: nothing is written
        bad_frames = (C.c_int32 * S)(*[int(nfr[0])] + [-1] * (S - 1))
        bad_op = (_abi.StepOp * (len(ops) + 1))()
        for o, t in zip(bad_op, ops + [(0, ne, 0, 0, 0, 0)]):
            o.kind, o.src, o.comp, o.dst, o.conv, o.mode = t
        plan_bad = _abi.StepPlan()
        C.memmove(C.addressof(plan_bad), C.addressof(plan), C.sizeof(plan))
        plan_bad.ops, plan_bad.n_ops = bad_op, len(ops) + 1
        for name, lib in libs.items():
            for pl, frs in ((plan, bad_frames), (plan_bad, frames)):
                rc, _sd, m2, x2, o2 = run(lib, pl, vals, chain, unc, frs, tmpl, mat, xf, off)
                assert rc == -1, (name, trial)
                np.testing.assert_array_equal(m2, mat)
                np.testing.assert_array_equal(x2, xf)
                np.testing.assert_array_equal(o2, off)
            sd = _abi.SceneDesc()
            assert lib.ffx_scene_step_h(plan, vals.ctypes.data, chain.ctypes.data, unc.ctypes.data, frames, tmpl, sd, None, xf.ctypes.data, off.ctypes.data, None, 1, None) == -1
    # ... and with a geometry block the call IS the re-fit (the oracle's, on host memory): two shapes, three frames each, against Geometry.update
    V, S = 5, 2
    pool = rng.standard_normal((S * 3 * V, 3)).astype(np.float32)
    tris = np.array([[0, 1, 2], [2, 3, 4], [0, 2, 4], [1, 3, 4]], np.int32)
    tri_shape = np.array([0, 0, 1, 1], np.int32)
    off0 = np.array([0, 3 * V], np.int32)
    ga, gb = oracle.Geometry(pool, tris, tri_shape, off0), oracle.Geometry(pool, tris, tri_shape, off0)
    chain, unc = rng.standard_normal((3, 16)).astype(np.float32), rng.standard_normal((3, 16)).astype(np.float32)
    for m in (chain, unc):
        m.reshape(3, 4, 4)[:, 3] = (0, 0, 0, 1)
    oarr = (_abi.StepOp * 2)()
    for o, t in zip(oarr, [(3, 2, 0, 0, 0, 0), (3, 1, 0, 1, 0, 1)]):
        o.kind, o.src, o.comp, o.dst, o.conv, o.mode = t
    base, fstride, nfr = off0.copy(), np.full(S, V, np.int32), np.full(S, 3, np.int32)
    plan = _abi.StepPlan()
    plan.ops, plan.n_ops, plan.n_shapes, plan.n_draws, plan.n_ents = oarr, 2, S, 0, 3
    plan.frame_base, plan.frame_stride, plan.n_frames = (a.ctypes.data_as(C.POINTER(C.c_int32)) for a in (base, fstride, nfr))
    geom = _abi.StepGeom()
    geom.bvh, geom.info = ga.blob.ctypes.data, C.pointer(ga.info)
    geom.src_verts, geom.tris, geom.tri_shape = ga.src_verts.ctypes.data, ga.tris.ctypes.data, ga.tri_shape.ctypes.data
    xf, off = np.tile(np.eye(4, dtype=np.float32).reshape(-1), (S, 1)), off0.copy()
    tmpl, sd = _abi.SceneDesc(), _abi.SceneDesc()
    assert libs["oracle"].ffx_scene_step_h(plan, None, chain.ctypes.data, unc.ctypes.data, (C.c_int32 * S)(2, 1), tmpl, sd, None, xf.ctypes.data, off.ctypes.data, geom, 1, None) == 0
    np.testing.assert_array_equal(off, [2 * V, 3 * V + V])
    gb.update(np.stack([unc[2], chain[1]]).reshape(S, 4, 4), vert_off=off)
    np.testing.assert_array_equal(ga.blob, gb.blob)
    geom.tris = None
    assert libs["oracle"].ffx_scene_step_h(plan, None, chain.ctypes.data, unc.ctypes.data, None, tmpl, sd, None, xf.ctypes.data, off.ctypes.data, geom, 1, None) == -1


def test_mitsuba_array_shims_carry_the_arithmetic_of_depth_py():
    """Round-4 advisor: the reference's own call sites compute with Mitsuba's arrays before and after the entry points the shims serve —
    fireflies/graphics/depth.py:61-69 (`pos //= spp`, `pos % w`, `pos // w`, `mi.Float(...)` of an array, `mi.Vector2f(x, y)`, `pos * scale`)
    and :84 (`result[~si.is_valid()] = 0`).  The same statements on fireflies_amd.mi's wrappers, against numpy."""
    import numpy as np
    import torch

    from fireflies_amd import mi

    W, H, spp = 7, 5, 3
    total = W * H * spp
    pos = mi.UInt32(torch.arange(total, dtype=torch.int32))
    pos //= spp
    scale = mi.Vector2f(1.0 / W, 1.0 / H)
    p2 = mi.Vector2f(mi.Float(pos % int(W)), mi.Float(pos // int(W)))
    got = (p2 * scale).numpy()
    idx = np.arange(total) // spp
    want = np.stack([(idx % W) / W, (idx // W) / H], -1).astype(np.float32)
    np.testing.assert_allclose(got, want, rtol=1e-6)
    assert isinstance(p2 * scale, mi.Vector2f) and isinstance(mi.Float(pos % int(W)), mi.Float32) and isinstance(mi.Float(2.5), float)
    # masked assignment with a negated validity mask
    result = mi.Float32(torch.arange(6, dtype=torch.float32) + 1.0)
    valid = torch.tensor([True, False, True, True, False, True])
    result[~valid] = 0
    assert result.numpy().tolist() == [1.0, 0.0, 3.0, 4.0, 0.0, 6.0]
    m = mi.UInt32(torch.tensor([1, 0, 3], dtype=torch.int32))
    assert ((m + 1) * 2 - 1).numpy().tolist() == [3, 1, 7] and (2 * m).numpy().tolist() == [2, 0, 6] and (-mi.Float32([1.0, -2.0])).numpy().tolist() == [-1.0, 2.0]
