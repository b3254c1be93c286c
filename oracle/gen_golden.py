#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE's own torch code.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where the reference is
mounted at /root/reference; it never runs on the GPU box (the reference does not
travel).  The output (.npz files under tests/golden/) is data: inputs and the outputs
the reference computed for them.  No reference source text is stored.

The reference cannot be imported as-is: `mitsuba`, `drjit`,
`pywavefront`, `geomdl`, `cv2`, `kornia` are not installed, so empty stub modules are
put into sys.modules and only the torch-only half of the package is exercised, always
on device="cpu".

Fixture ids follow SURVEY.md §8(c): g1..g9.
"""
import os
import sys
import types
import random

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)
    for name in ["mitsuba", "drjit", "pywavefront", "geomdl", "cv2", "kornia"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["geomdl"].NURBS = types.SimpleNamespace(Curve=object)
    import fireflies  # noqa: F401
    import fireflies.graphics.rasterization as R
    import fireflies.utils.math as M
    import fireflies.sampling as S
    import fireflies.entity as E
    import fireflies.projection as P

    return R, M, S, E, P


def perspective_projection(W, H, fov_x_deg, near, far):
    """Mitsuba 3 `mi.perspective_projection` for crop == full film [EXT, SURVEY App. A].
    Stored in the fixture so that nothing depends on Mitsuba being importable."""
    aspect = W / H
    c = 1.0 / np.tan(np.deg2rad(fov_x_deg) * 0.5)
    K = np.array(
        [
            [-0.5 * c, 0.0, 0.5, 0.0],
            [0.0, -0.5 * aspect * c, 0.5, 0.0],
            [0.0, 0.0, far / (far - near), -near * far / (far - near)],
            [0.0, 0.0, 1.0, 0.0],
        ],
        dtype=np.float32,
    )
    return K


def main():
    os.makedirs(OUT, exist_ok=True)
    R, M, S, E, P = import_reference()
    cpu = "cpu"
    f32 = torch.float32

    # ---------------------------------------------------------------- g1
    g1 = {}
    for n in (8, 16, 18, 32):
        g1[f"rays_{n}"] = P.Laser.generate_uniform_rays(0.0275, n, n, device=cpu).numpy()
    g1["rays_8_wide"] = P.Laser.generate_uniform_rays(0.0275 * 18 / 8, 8, 8, device=cpu).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_uniform_rays.npz"), **g1)

    # ---------------------------------------------------------------- g2 / g9
    K = torch.from_numpy(perspective_projection(500, 500, 30.0, 0.01, 100.0))
    tr = E.Transformable("projector", cpu)
    g2 = {"K": K.numpy()}
    for n in (8, 18):
        rays = P.Laser.generate_uniform_rays(0.0275 * 18 / n, n, n, device=cpu)
        laser = P.Laser(tr, rays.clone(), K, 30.0, 0.01, 100.0, device=cpu)
        ndc = laser.projectRaysToNDC()
        back = laser.projectNDCPointsToWorld(ndc)
        g2[f"rays_{n}"] = rays.numpy()
        g2[f"ndc_{n}"] = ndc.numpy()
        g2[f"back_{n}"] = back.numpy()
        # autograd through projectRaysToNDC (K1 backward)
        laser._rays = rays.clone().requires_grad_(True)
        ndc2 = laser.projectRaysToNDC()
        w = torch.linspace(-1.0, 1.0, ndc2.numel()).reshape(ndc2.shape)
        (ndc2 * w).sum().backward()
        g2[f"gw_{n}"] = w.numpy()
        g2[f"grays_{n}"] = laser._rays.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_projection.npz"), **g2)

    # g9: clamp_to_fov with a wide pattern that leaves the frustum
    rays = P.Laser.generate_uniform_rays(0.06, 12, 12, device=cpu)
    laser = P.Laser(tr, rays.clone(), K, 30.0, 0.01, 100.0, device=cpu)
    before = laser._rays.clone().numpy()
    ndc_before = laser.projectRaysToNDC().numpy()
    laser.clamp_to_fov()
    g9 = {
        "K": K.numpy(),
        "rays_before": before,
        "ndc_before": ndc_before,
        "rays_after": laser._rays.numpy(),
        "ndc_after": laser.projectRaysToNDC().numpy(),
    }
    laser.normalize_rays()
    g9["rays_after_normalize"] = laser._rays.numpy()
    np.savez_compressed(os.path.join(OUT, "g9_clamp_to_fov.npz"), **g9)

    # ---------------------------------------------------------------- g3
    # rasterize_points + sum / softor, forward and autograd gradient wrt points
    g3 = {}
    cases = []
    torch.manual_seed(1234)
    # (name, N, size=(size0,size1), sigma)
    for name, N, size, sigma in [
        ("a", 1, (16, 16), 4.0),
        ("b", 4, (16, 16), 10.0),
        ("c", 4, (8, 16), 4.0),
        ("d", 64, (64, 48), 10.0),
        ("e", 64, (64, 48), 100.0),
        ("f", 16, (100, 100), 10.0),
        ("g", 7, (33, 21), 4.0),
    ]:
        pts = torch.rand(N, 2, dtype=f32)
        if N >= 4:
            pts[0] = torch.tensor([0.0, 0.0])  # on the border
            pts[1] = torch.tensor([1.0, 1.0])  # on the far border
            pts[2] = torch.tensor([-0.1, 0.5])  # outside
            pts[3] = torch.tensor([0.5, 1.2])  # outside
        cases.append((name, pts, size, sigma))
    g3["survey_anchor_pts"] = None
    for name, pts, size, sigma in cases:
        tsize = torch.tensor(list(size))
        for mode in ("sum", "softor"):
            p = pts.clone().requires_grad_(True)
            dense = R.rasterize_points(p, sigma, tsize, device=cpu)
            red = R.sum(dense) if mode == "sum" else R.softor(dense)
            w = torch.sin(torch.arange(red.numel(), dtype=f32) * 0.37).reshape(red.shape)
            (red * w).sum().backward()
            if mode == "sum":
                g3[f"{name}_dense"] = dense.detach().numpy()
                g3[f"{name}_w"] = w.numpy()
            g3[f"{name}_{mode}"] = red.detach().numpy()
            g3[f"{name}_{mode}_gpts"] = p.grad.numpy()
        g3[f"{name}_pts"] = pts.numpy()
        g3[f"{name}_size"] = np.array(size, dtype=np.int64)
        g3[f"{name}_sigma"] = np.float32(sigma)
        # dense backward with an arbitrary upstream gradient on [N,H,W]
        p = pts.clone().requires_grad_(True)
        dense = R.rasterize_points(p, sigma, tsize, device=cpu)
        wd = torch.cos(torch.arange(dense.numel(), dtype=f32) * 0.11).reshape(dense.shape)
        (dense * wd).sum().backward()
        g3[f"{name}_dense_w"] = wd.numpy()
        g3[f"{name}_dense_gpts"] = p.grad.numpy()
    del g3["survey_anchor_pts"]
    g3["case_names"] = np.array([c[0] for c in cases])
    # SURVEY App. C anchors
    a = R.rasterize_points(torch.tensor([[0.25, 0.75]]), 4.0, torch.tensor([8, 16]), device=cpu)
    g3["anchor1"] = a.numpy()
    torch.manual_seed(0)
    pa = torch.rand(4, 2)
    g3["anchor2_pts"] = pa.numpy()
    g3["anchor2"] = R.rasterize_points(pa, 10.0, torch.tensor([16, 16]), device=cpu).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_rasterize_points.npz"), **g3)

    # one full-size forward (500x500, N=64, sigma=10): keep only the reduced textures
    torch.manual_seed(77)
    pts = torch.rand(64, 2, dtype=f32) * 0.9 + 0.05
    dense = R.rasterize_points(pts, 10.0, torch.tensor([500, 500]), device=cpu)
    g3b = {
        "pts": pts.numpy(),
        "sum": R.sum(dense).numpy().astype(np.float16 if False else np.float32),
        "softor": R.softor(dense).numpy(),
    }
    np.savez_compressed(os.path.join(OUT, "g3_full_500.npz"), **g3b)

    # ---------------------------------------------------------------- g4 baked variants
    g4 = {}
    torch.manual_seed(5)
    size = torch.tensor([100, 100])
    sig2 = torch.tensor(10.0**2)
    pts_in = torch.rand(10, 2) * 0.5 + 0.25  # interior
    pts_bd = torch.rand(12, 2)  # some near the border
    pts_bd[0] = torch.tensor([0.01, 0.5])
    pts_bd[1] = torch.tensor([0.5, 0.01])
    pts_bd[2] = torch.tensor([0.995, 0.5])
    pts_bd[3] = torch.tensor([0.5, 0.995])
    pts_bd[4] = torch.tensor([0.02, 0.03])
    pts_bd[5] = torch.tensor([0.97, 0.98])
    for tag, pts in (("in", pts_in), ("bd", pts_bd)):
        g4[f"{tag}_pts"] = pts.numpy()
        g4[f"{tag}_baked_sum"] = R.baked_sum(pts, sig2, size, 4, device=cpu).numpy()
        g4[f"{tag}_baked_sum_2"] = R.baked_sum_2(pts, sig2, size, 4, device=cpu).numpy()
        g4[f"{tag}_baked_softor"] = R.baked_softor(pts, sig2, size, 5, device=cpu).numpy()
        try:
            g4[f"{tag}_baked_softor_2"] = R.baked_softor_2(pts, sig2, size, 5, device=cpu).numpy()
        except Exception as e:  # reference border code may raise on shape mismatch
            g4[f"{tag}_baked_softor_2_error"] = np.array(str(type(e).__name__))
        dense = R.rasterize_points(pts, 100.0, size, device=cpu)
        g4[f"{tag}_dense_sum"] = R.sum(dense).numpy()
        g4[f"{tag}_dense_softor"] = R.softor(dense).numpy()
    # non-square texture (orientation check)
    size2 = torch.tensor([60, 40])
    pts = torch.rand(6, 2) * 0.4 + 0.3
    g4["ns_pts"] = pts.numpy()
    g4["ns_size"] = size2.numpy()
    g4["ns_baked_sum"] = R.baked_sum(pts, torch.tensor(16.0), size2, 4, device=cpu).numpy()
    g4["ns_baked_softor"] = R.baked_softor(pts, torch.tensor(16.0), size2, 5, device=cpu).numpy()
    g4["ns_dense_sum"] = R.sum(R.rasterize_points(pts, 16.0, size2, device=cpu)).numpy()
    # gradient of baked_sum wrt points (interior)
    p = pts_in.clone().requires_grad_(True)
    out = R.baked_sum(p, sig2, size, 4, device=cpu)
    w = torch.sin(torch.arange(out.numel(), dtype=f32) * 0.21).reshape(out.shape)
    (out * w).sum().backward()
    g4["in_baked_sum_w"] = w.numpy()
    g4["in_baked_sum_gpts"] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g4_baked.npz"), **g4)

    # ---------------------------------------------------------------- g5
    g5 = {}
    torch.manual_seed(9)
    pts3 = torch.rand(5, 3)
    size = torch.tensor([24, 20])
    p = pts3.clone().requires_grad_(True)
    rd = R.rasterize_depth(p[:, 0:2], p[:, 2:3], 6.0, size, device=cpu)
    w = torch.sin(torch.arange(rd.numel(), dtype=f32) * 0.3).reshape(rd.shape)
    (rd * w).sum().backward()
    g5["depth_pts"] = pts3.numpy()
    g5["depth_size"] = size.numpy()
    g5["depth_out"] = rd.detach().numpy()
    g5["depth_w"] = w.numpy()
    g5["depth_gpts"] = p.grad.numpy()
    lines = torch.rand(4, 2, 2)
    lsize = torch.tensor([20, 20])  # reference's meshgrid only consistent for square
    lo = R.rasterize_lines(lines.clone(), 3.0, lsize, device=cpu)
    g5["lines_in"] = lines.numpy()
    g5["lines_size"] = lsize.numpy()
    g5["lines_out"] = lo.numpy()
    lsize2 = torch.tensor([24, 16])
    lo2 = R.rasterize_lines(lines.clone(), 3.0, lsize2, device=cpu)
    g5["lines_size2"] = lsize2.numpy()
    g5["lines_out2"] = lo2.numpy()
    # subsampled_point_raster: defaults to device cuda inside -> call pieces on cpu
    sub = []
    for i in range(3):
        r = R.rasterize_depth(pts3[:, 0:2], pts3[:, 2:3], 6.0, torch.tensor([32, 32]) // 2**i, device=cpu)
        sub.append(R.softor(r, keepdim=True).numpy())
    for i, s_ in enumerate(sub):
        g5[f"subsampled_{i}"] = s_
    np.savez_compressed(os.path.join(OUT, "g5_depth_lines.npz"), **g5)

    # ---------------------------------------------------------------- g6
    g6 = {}
    torch.manual_seed(3)
    pts = torch.randn(9, 3)
    T = torch.eye(4)
    T[:3, :3] = M.getYawTransform(0.3, cpu) @ M.getPitchTransform(-0.7, cpu) @ M.getRollTransform(1.1, cpu)
    T[:3, 3] = torch.tensor([0.5, -1.0, 2.0])
    g6["pts"] = pts.numpy()
    g6["T"] = T.numpy()
    g6["transform_points"] = M.transform_points(pts, T).numpy()
    g6["transform_points_K"] = M.transform_points(pts, K).numpy()
    g6["transform_directions"] = M.transform_directions(pts, T).numpy()
    g6["toMat4x4"] = M.toMat4x4(T[:3, :3].clone()).numpy()
    g6["toMat4x4_noone"] = M.toMat4x4(T[:3, :3].clone(), addOne=False).numpy()
    for nm in ("getYawTransform", "getPitchTransform", "getRollTransform", "getXTransform", "getYTransform", "getZTransform"):
        g6[nm] = getattr(M, nm)(0.4, cpu).numpy()
    torch.manual_seed(11)
    a = torch.tensor([0.0, -1.0, 2.0])
    b = torch.tensor([1.0, 1.0, 2.0])
    g6["rbt_a"], g6["rbt_b"] = a.numpy(), b.numpy()
    g6["randomBetweenTensors"] = M.randomBetweenTensors(a, b).numpy()
    t = torch.tensor([[1.0, 5.0], [3.0, -1.0]])
    g6["normalize_in"] = t.numpy()
    g6["normalize"] = M.normalize(t).numpy()
    v1, v2 = torch.tensor([1.0, 0.2, 0.0]), torch.tensor([0.0, 1.0, 0.5])
    g6["rmfv_v1"], g6["rmfv_v2"] = v1.numpy(), v2.numpy()
    g6["rotation_matrix_from_vectors"] = M.rotation_matrix_from_vectors(v1, v2).numpy()
    np.savez_compressed(os.path.join(OUT, "g6_math.npz"), **g6)

    # ---------------------------------------------------------------- g7
    g7 = {}
    verts = torch.tensor(
        [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 2.0, 0.0], [0.0, 0.0, 3.0], [1.0, 1.0, 1.0]]
    )
    centroid = verts.sum(dim=0, keepdim=True) / verts.shape[0]
    g7["verts"] = verts.numpy()
    for s in (0, 1, 2):
        torch.manual_seed(s)
        t = E.Transformable("x", cpu)
        t.rotate_z(-1, 1)
        t.translate_x(-0.5, 0.5)
        t.train()
        t.randomize()
        g7[f"tr_world_{s}"] = t.world().numpy()

        torch.manual_seed(s)
        t = E.Transformable("y", cpu)
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
        g7[f"tr_full_world_{s}"] = t.world().numpy()
        g7[f"tr_full_fkey_{s}"] = t.get_randomized_float_attributes()["fkey"].numpy()
        g7[f"tr_full_vkey_{s}"] = t.get_randomized_vec3_attributes()["vkey"].numpy()

        torch.manual_seed(s)
        m = E.Mesh("mesh", verts - centroid, cpu)
        m.set_centroid(centroid)
        m.scale_x(0.5, 2.0)
        m.scale_z(1.0, 3.0)
        m.rotate_y(-0.25, 0.25)
        m.translate_y(-0.05, 0.05)
        m.train()
        m.randomize()
        g7[f"mesh_world_{s}"] = m.world().numpy()
        g7[f"mesh_verts_{s}"] = m.get_randomized_vertices().numpy()

        # parent / child chain
        torch.manual_seed(s)
        parent = E.Mesh("parent", verts - centroid, cpu)
        parent.set_centroid(centroid)
        parent.rotate_x(-0.5, 0.5)
        child = E.Mesh("child", (verts - centroid) * 0.5, cpu)
        child.translate_z(0.1, 0.9)
        child.setParent(parent)
        parent.train()
        child.train()
        parent.randomize()
        child.randomize()
        g7[f"pc_parent_world_{s}"] = parent.world().numpy()
        g7[f"pc_child_world_{s}"] = child.world().numpy()
        g7[f"pc_child_verts_{s}"] = child.get_randomized_vertices().numpy()

        # animation function (uniform float time)
        torch.manual_seed(s)
        am = E.Mesh("anim", verts - centroid, cpu)
        am.add_animation_func(lambda v, t_: v * (1.0 + t_), 0.0, 1.0)
        am.rotate_z(-0.2, 0.2)
        am.train()
        am.randomize()
        g7[f"anim_verts_{s}"] = am.get_randomized_vertices().numpy()
    np.savez_compressed(os.path.join(OUT, "g7_randomize.npz"), **g7)

    # ---------------------------------------------------------------- g8
    g8 = {}
    smp = S.UniformSampler(0.0, 0.05, device=cpu)
    smp.eval()
    g8["uniform_scalar_eval"] = np.array([float(smp.sample()) for _ in range(12)], dtype=np.float32)
    g8["uniform_scalar_min_after"] = smp.get_min().numpy().copy()
    smp = S.UniformSampler(torch.tensor([0.0, 1.0, 2.0]), torch.tensor([0.03, 1.03, 2.03]), device=cpu)
    smp.eval()
    g8["uniform_vec3_eval"] = np.stack([smp.sample().clone().numpy() for _ in range(8)])
    smp = S.UniformSampler(torch.tensor([0.0, 1.0, 2.0]), torch.tensor([0.03, 1.0, 2.0]), device=cpu)
    smp.eval()
    g8["uniform_vec3_degenerate_eval"] = np.stack([smp.sample().clone().numpy() for _ in range(8)])
    smp = S.UniformSampler(torch.tensor([1.0, 1.0, 1.0]), torch.tensor([1.0, 1.0, 1.0]), device=cpu)
    smp.eval()
    g8["uniform_const_eval"] = np.stack([smp.sample().clone().numpy() for _ in range(3)])
    an = S.AnimationSampler(0, 1, 0, 1, device=cpu)
    an.set_eval_interval(0, 5)
    an.eval()
    g8["animation_eval"] = np.array([an.sample() for _ in range(14)], dtype=np.int64)
    an = S.AnimationSampler(0, 7, 0, 3, device=cpu)
    an.train()
    random.seed(42)
    g8["animation_train_seed42"] = np.array([an.sample() for _ in range(20)], dtype=np.int64)
    torch.manual_seed(21)
    sv = S.UniformScalarToVec3Sampler(1.0, 20.0, device=cpu)
    sv.train()
    g8["scalar_to_vec3_train_seed21"] = np.stack([sv.sample().numpy() for _ in range(4)])
    sv.eval()
    g8["scalar_to_vec3_eval"] = np.stack([sv.sample().numpy() for _ in range(4)])
    torch.manual_seed(22)
    gs = S.GaussianSampler(torch.tensor([0.0]), torch.tensor([1.0]), torch.tensor([0.5, 0.5]), torch.tensor([0.1, 0.2]), device=cpu)
    gs.train()
    g8["gaussian_train_seed22"] = np.stack([gs.sample().numpy() for _ in range(4)])
    np.savez_compressed(os.path.join(OUT, "g8_samplers.npz"), **g8)

    print("golden fixtures written to", os.path.abspath(OUT))
    for f in sorted(os.listdir(OUT)):
        print("  ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
