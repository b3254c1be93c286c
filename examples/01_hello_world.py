"""BASELINE configs[0] counterpart of the reference's examples/01_hello_world.py: load a scene,
randomise a mesh, render.  256x256, 16 spp."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fireflies_amd import mi, scenes  # noqa: E402

mi.set_variant("cuda_ad_rgb")
import fireflies_amd as ff  # noqa: E402

if __name__ == "__main__":
    mi_scene = mi.load_scene_data(scenes.hello_world(256, 256))
    mi_params = mi.traverse(mi_scene)
    ff_scene = ff.Scene(mi_params)
    mesh = ff_scene.mesh("mesh-Cube")
    mesh.rotate_y(-0.5, 0.5)  # about Z in the reference's convention (entity/base.py:194-207)
    mesh.translate_x(-0.5, 0.5)
    ff_scene.train()
    ff_scene.randomize()
    render = mi.render(mi_scene, spp=16)
    print("rendered", tuple(render.torch().shape), "max", float(render.torch().max()))
