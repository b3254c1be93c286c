"""Ready-made workloads = BASELINE.json configs on the procedural scenes.  Used by
bench.py, __graft_entry__.smoke(), the examples and the GPU tests, so that they all measure and
check the same thing."""
from dataclasses import dataclass

import torch

from . import mi, scenes
from .projection import Laser
from .scene import Scene


@dataclass
class Workload:
    data: scenes.SceneData
    mi_scene: mi.Scene
    params: mi.SceneParameters
    ff_scene: Scene
    laser: Laser
    K_projector: torch.Tensor
    sigma: float
    tex_size: tuple


def vocalfold(device="cuda", width=512, height=512, tex=500, grid=16, frames=50, n_fold=96, tube=(64, 128), shadows=True, randomize=True,
              entity_device=None, principled=True):
    """configs[1]/[2] of BASELINE.json: animated vocal-fold scene (53,248 triangles at the default
    detail), `grid` x `grid` point laser, camera width x height, projector texture tex x tex.
    Randomisation ranges are those of examples/vocalfold_scene.py:73-92.
    `entity_device` is the device argument of ff.Scene (where the samplers draw): the reference
    default is the GPU; "cpu" draws from torch's CPU generator and avoids one device sync per draw.
    `principled`: the material is Mitsuba's principled BSDF, whose `specular` the reference randomises (False: diffuse)."""
    edev = device if entity_device is None else entity_device
    data = scenes.vocalfold(width=width, height=height, tex=tex, frames=frames, n_fold=n_fold, tube=tube, principled=principled)
    mi_scene = mi.load_scene_data(data, device=device, shadows=shadows)
    params = mi.traverse(mi_scene)
    ff_scene = Scene(params, device=edev)
    if randomize:
        larynx, fold = ff_scene.mesh("mesh-Larynx"), ff_scene.mesh("mesh-VocalFold")
        larynx.scale_x(0.8, 1.2)
        larynx.rotate_y(-0.1, 0.1)
        fold.scale_x(0.5, 2.0)
        fold.rotate_y(-0.25, 0.25)
        n_train = max(1, (frames * 4) // 5)
        fold.set_pool_animation(n_train, max(1, frames - n_train))  # 40 train / 10 eval, cf. main.py:84-85
        from .sampling import UniformScalarToVec3Sampler

        ff_scene.light("emit-Spot").add_vec3_sampler("intensity.value", UniformScalarToVec3Sampler(1.0, 20.0, device=edev))
        mat = ff_scene.material("mat-Default OBJ")
        mat.add_vec3_key("brdf_0.base_color.value", torch.tensor([0.8, 0.14, 0.34], device=edev), torch.tensor([0.85, 0.5, 0.44], device=edev))
        mat.add_float_key("brdf_0.specular", 0.0, 0.75)
    ff_scene.train()
    proj = mi_scene.sensors()[1]
    x_fov, near, far = params[proj.id() + ".x_fov"], params[proj.id() + ".near_clip"], params[proj.id() + ".far_clip"]
    K = mi.perspective_projection(proj.film().size(), proj.film().crop_size(), proj.film().crop_offset(), x_fov, near, far).matrix.torch()[0]
    rays = Laser.generate_uniform_rays(0.0275 * 18 / grid, grid, grid, device=device)
    laser = Laser(ff_scene._projector, rays, K, x_fov, near, far, device=device)
    return Workload(data, mi_scene, params, ff_scene, laser, K, 10.0, (tex, tex))


def colon(device="cuda", width=1024, height=1024, tex=1024, grid=32, shadows=True, randomize=True, entity_device=None, principled=True,
          n_around=256, n_along=1024):
    """configs[4] of BASELINE.json: colon-endoscopy scene (524,288 triangles), 1024x1024, 1024-point
    pattern (32 x 32), texture 1024^2; render with spp = 256 and fp16=True for the full configuration."""
    edev = device if entity_device is None else entity_device
    data = scenes.colon(width=width, height=height, tex=tex, principled=principled, n_around=n_around, n_along=n_along)
    mi_scene = mi.load_scene_data(data, device=device, shadows=shadows)
    params = mi.traverse(mi_scene)
    ff_scene = Scene(params, device=edev)
    if randomize:
        mesh = ff_scene.mesh("mesh-Colon")
        mesh.scale_x(0.9, 1.1)
        mesh.rotate_z(-0.05, 0.05)
        from .sampling import UniformScalarToVec3Sampler

        ff_scene.light("emit-Spot").add_vec3_sampler("intensity.value", UniformScalarToVec3Sampler(2.0, 10.0, device=edev))
        if principled:  # the mucosa randomisation of main.py:97-107 (spec_trans scales the diffuse lobe only, include/ffx.h)
            mat = ff_scene.material("mat-Mucosa")
            mat.add_float_key("brdf_0.clearcoat.value", 0.0, 1.0)
            mat.add_float_key("brdf_0.clearcoat_gloss.value", 0.0, 1.0)
            mat.add_float_key("brdf_0.metallic.value", 0.0, 0.5)
            mat.add_float_key("brdf_0.specular", 0.0, 1.0)
            mat.add_float_key("brdf_0.roughness.value", 0.0, 1.0)
            mat.add_float_key("brdf_0.anisotropic.value", 0.0, 1.0)
            mat.add_float_key("brdf_0.sheen.value", 0.0, 0.5)
            mat.add_float_key("brdf_0.spec_trans.value", 0.0, 0.4)
            mat.add_float_key("brdf_0.flatness.value", 0.0, 1.0)
    ff_scene.train()
    proj = mi_scene.sensors()[1]
    x_fov, near, far = params[proj.id() + ".x_fov"], params[proj.id() + ".near_clip"], params[proj.id() + ".far_clip"]
    K = mi.perspective_projection(proj.film().size(), proj.film().crop_size(), proj.film().crop_offset(), x_fov, near, far).matrix.torch()[0]
    rays = Laser.generate_uniform_rays(0.0275 * 18 / grid * 1.8, grid, grid, device=device)
    laser = Laser(ff_scene._projector, rays, K, x_fov, near, far, device=device)
    return Workload(data, mi_scene, params, ff_scene, laser, K, 10.0, (tex, tex))


def build_texture(wl: Workload, reduce="sum", blur=True):
    """the laser texture as in examples/vocalfold_scene.py:56-67, kept 1-channel (green weight
    lives in the scene description) and on the device."""
    from . import functional as Fn

    t = wl.laser.generateTexture(wl.sigma, wl.tex_size, reduce=reduce)
    return Fn.gaussian_blur(t, 5, 3.0) if blur else t
