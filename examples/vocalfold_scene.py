"""Counterpart of the reference's examples/vocalfold_scene.py with the imports swapped: laser
texture from a point pattern, randomised vocal-fold scene, render loop.  The scene comes from
fireflies_amd.scenes (the reference's XML / OBJ assets are not distributed)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fireflies_amd import mi, scenes  # instead of `import mitsuba as mi`

mi.set_variant("cuda_ad_rgb")

import fireflies_amd as fireflies  # noqa: E402  instead of `import fireflies`
import fireflies_amd.sampling  # noqa: E402,F401
from fireflies_amd import functional as Fn  # noqa: E402


def render_to_uint8(render):
    img = torch.clamp(render.torch(), 0, 1)[:, :, [2, 1, 0]].cpu().numpy()
    return (img * 255).astype(np.uint8)


if __name__ == "__main__":
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mitsuba_scene = mi.load_scene_data(scenes.vocalfold())  # reference: mi.load_file("…/vocalfold.xml")
    mitsuba_params = mi.traverse(mitsuba_scene)
    ff_scene = fireflies.Scene(mitsuba_params)

    projector_sensor = mitsuba_scene.sensors()[1]
    x_fov = mitsuba_params["PerspectiveCamera_1.x_fov"]
    near_clip = mitsuba_params["PerspectiveCamera_1.near_clip"]
    far_clip = mitsuba_params["PerspectiveCamera_1.far_clip"]
    K_PROJECTOR = mi.perspective_projection(
        projector_sensor.film().size(), projector_sensor.film().crop_size(), projector_sensor.film().crop_offset(), x_fov, near_clip, far_clip
    ).matrix.torch()[0]

    laser_rays = fireflies.projection.Laser.generate_uniform_rays(0.0275, 18, 18, device=ff_scene.device())
    laser = fireflies.projection.Laser(ff_scene._projector, laser_rays, K_PROJECTOR, x_fov, near_clip, far_clip, device=ff_scene.device())
    texture = laser.generateTexture(10.0, torch.tensor([500, 500], device=ff_scene.device()))  # [N,500,500] on the device
    texture = texture.sum(dim=0)
    texture = Fn.gaussian_blur(texture, 5, 3.0)  # reference: kornia.filters.gaussian_blur2d(…, (5, 5), (3, 3))
    texture = torch.stack([torch.zeros_like(texture), texture, torch.zeros_like(texture)])
    texture = torch.movedim(texture, 0, -1).contiguous()
    mitsuba_params["tex.data"] = mi.TensorXf(texture)  # stays on the device (reference: .cpu().numpy())

    vocalfold_mesh = ff_scene.mesh("mesh-VocalFold")
    larynx_mesh = ff_scene.mesh("mesh-Larynx")
    larynx_mesh.scale_x(0.8, 1.2)
    larynx_mesh.rotate_y(-0.1, 0.1)
    vocalfold_mesh.scale_x(0.5, 2.0)
    vocalfold_mesh.rotate_y(-0.25, 0.25)
    vocalfold_mesh.set_pool_animation(40, 10)  # reference: add_train/eval_animation_from_obj(dir)

    material = ff_scene.material("mat-Default OBJ")
    light = ff_scene.light("emit-Spot")
    light.add_vec3_sampler("intensity.value", fireflies.sampling.UniformScalarToVec3Sampler(1.0, 20.0, device=ff_scene.device()))
    material.add_vec3_key("brdf_0.base_color.value", torch.tensor([0.8, 0.14, 0.34], device=ff_scene.device()),
                          torch.tensor([0.85, 0.5, 0.44], device=ff_scene.device()))
    material.add_float_key("brdf_0.specular", 0.0, 0.75)

    ff_scene.train()
    os.makedirs("vf_renderings", exist_ok=True)
    for i in range(n_images):
        ff_scene.randomize()
        render = mi.render(mitsuba_scene, spp=100)
        img = render_to_uint8(render)
        try:
            from PIL import Image

            Image.fromarray(img[:, :, ::-1]).save("vf_renderings/{0:05d}.png".format(i))
        except ImportError:
            np.save("vf_renderings/{0:05d}.npy".format(i), img)
    print("wrote", n_images, "renderings to vf_renderings/")
