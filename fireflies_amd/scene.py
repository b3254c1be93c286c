"""Scene — fireflies/scene.py re-stated over `fireflies_amd.mi` parameters.

Same construction rules (key classification by substring, first match wins, keys visited in
sorted order, scene.py:92-116), same entity API (`mesh(name)`, `light(name)`, `material(name)`,
`train()/eval()/randomize()`), same randomisation order (meshes, lights, materials — parents
before children — then camera, then projector; scene.py:344-371).

What differs is where the work happens.  The reference transforms every vertex in torch,
round-trips it into `mi.Float32` and lets `params.update()` rebuild Mitsuba's acceleration
structure (scene.py:243-251,384).  Here `update_meshes` hands (world matrix, animation frame) to
the parameter object and `params.update()` issues ONE fused device pass (vertex transform +
triangle records + BVH refit; ffx_scene_update).
"""
from typing import List

import torch

from . import emitter, entity, material, mi


class Scene:
    MESH_KEYS = ["mesh", "ply"]
    CAM_KEYS = ["camera", "perspective", "perspectivecamera"]
    PROJ_KEYS = ["projector"]
    MAT_KEYS = ["mat", "bsdf"]
    LIGHT_KEYS = ["light", "spot"]
    TEX_KEYS = ["tex"]

    def __init__(self, mitsuba_params, device=torch.device("cuda")):
        self._meshes = []
        self._projector = None
        self._camera = None
        self._lights = []
        self._curves = []
        self._materials = []
        self._transformables = []
        self._device = device
        self._mitsuba_params = mitsuba_params
        self.init_from_params(self._mitsuba_params)

    def device(self):
        return self._device

    # ------------------------------------------------------------------ lookup helpers
    @staticmethod
    def _by_name(items, name):
        for it in items:
            if it.name() == name:
                return it
        return None

    def mesh_at(self, index: int):
        return self._meshes[index]

    def meshes(self):
        return self._meshes

    def get_mesh(self, name: str):
        return self._by_name(self._meshes, name)

    def mesh(self, name: str):
        return self.get_mesh(name)

    def light_at(self, index: int):
        return self._lights[index]

    def lights(self):
        return self._lights

    def get_light(self, name: str):
        return self._by_name(self._lights, name)

    def light(self, name: str):
        return self.get_light(name)

    def material_at(self, index: int):
        return self._materials[index]

    def materials(self):
        return self._materials

    def get_material(self, name: str):
        return self._by_name(self._materials, name)

    def material(self, name: str):
        return self.get_material(name)

    # ------------------------------------------------------------------ construction
    def init_from_params(self, mitsuba_params) -> None:
        roots = sorted({key.split(".")[0] for key in mitsuba_params.keys()})
        table = [(self.MESH_KEYS, self.load_mesh), (self.CAM_KEYS, self.load_camera), (self.PROJ_KEYS, self.load_projector),
                 (self.LIGHT_KEYS, self.load_light), (self.MAT_KEYS, self.load_material)]
        for root in roots:
            low = root.lower()
            for keys, loader in table:
                if any(k.lower() in low for k in keys):
                    loader(root)
                    break

    def _to_world_of(self, base_key):
        return self._mitsuba_params[base_key + ".to_world"].matrix.torch().squeeze().to(self._device)

    def load_mesh(self, base_key: str):
        flat = self._mitsuba_params[base_key + ".vertex_positions"]
        flat = flat.torch() if hasattr(flat, "torch") else torch.as_tensor(flat)
        vertices = flat.to(self._device).reshape(-1, 3)
        centroid = vertices.sum(dim=0, keepdim=True) / vertices.shape[0]
        m = entity.Mesh(base_key, vertices - centroid, self._device)
        m.set_centroid(centroid)
        self._meshes.append(m)

    def _load_posed(self, base_key):
        t = entity.Transformable(base_key, self._device)
        t.set_world(self._to_world_of(base_key))
        t.set_randomizable(False)
        return t

    def load_camera(self, base_key: str) -> None:
        # every camera-like key lands here; like the reference the LAST one in sorted order wins
        # (scene.py:134-140; main.py:32 renames it by hand)
        self._camera = self._load_posed(base_key)

    def load_projector(self, base_key: str) -> None:
        self._projector = self._load_posed(base_key)

    def _harvest_attributes(self, ent, base_key):
        for key in [k for k in self._mitsuba_params.keys() if base_key in k]:
            rest = ".".join(key.split(".")[1:])
            value = self._mitsuba_params[key]
            if isinstance(value, (mi.Transform4f, mi.ScalarTransform3f)):
                continue
            if isinstance(value, (mi.Float, float)):
                ent.add_float_key(rest, value, value)
            elif hasattr(value, "__len__") and len(value) == 3:
                v = value.torch().squeeze().to(self._device) if hasattr(value, "torch") else torch.as_tensor(value, dtype=torch.float32, device=self._device)
                ent.add_vec3_key(rest, v, v)

    def load_light(self, base_key: str) -> None:
        light = emitter.Light(base_key, device=self._device)
        if base_key + ".to_world" in self._mitsuba_params.keys():
            light.set_world(self._to_world_of(base_key))
        self._harvest_attributes(light, base_key)
        light.set_randomizable(False)
        self._lights.append(light)

    def load_material(self, base_key: str) -> None:
        mat = material.Material(base_key, device=self._device)
        self._harvest_attributes(mat, base_key)
        mat.set_randomizable(False)
        self._materials.append(mat)

    def load_curve(self, path: str, name: str = "Curve") -> None:
        raise NotImplementedError("NURBS camera paths (fireflies/entity/curve.py) are outside the hot path; see DESIGN.md §7")

    # ------------------------------------------------------------------ mode switches
    def _everything(self):
        yield from self._meshes
        yield from self._lights
        yield from self._materials
        if self._camera is not None:
            yield self._camera
        if self._projector is not None:
            yield self._projector

    def train(self) -> None:
        for e in self._everything():
            e.train()

    def eval(self) -> None:
        for e in self._everything():
            e.eval()

    # ------------------------------------------------------------------ push to parameters
    def update_meshes(self) -> None:
        p = self._mitsuba_params
        fast = hasattr(p, "set_mesh_pose")
        for mesh in self._meshes:
            if not mesh.randomizable():
                continue
            if not fast:  # foreign parameter object: reference behaviour (scene.py:243-251)
                p[mesh.name() + ".vertex_positions"] = mi.Float32(mesh.get_randomized_vertices().flatten())
                continue
            pick = mesh.sample_animation_index()
            world = mesh._world_host()
            uncentre = torch.eye(4)
            uncentre[0:3, 3] = -mesh._centroid_mat[0:3, 3]
            if pick is None:
                p.set_mesh_pose(mesh.name(), world @ uncentre, frame=0)
            elif pick[0] == "func":
                p.set_mesh_pose(mesh.name(), world, vertices=mesh._animation_func(mesh._vertices, pick[2]))
            else:
                pool = getattr(mesh, "_pool_frames", None)
                if pool is not None:  # frames already resident in the device pool, stored like the rest pose
                    p.set_mesh_pose(mesh.name(), world @ uncentre, frame=pool[pick[1]][0] + pick[2])
                else:
                    stack = mesh._anim_data_train if pick[1] == "train" else mesh._anim_data_eval
                    p.set_mesh_pose(mesh.name(), world, vertices=stack[pick[2]])

    def _write_attributes(self, ent) -> None:
        p = self._mitsuba_params
        for key, value in ent.get_randomized_float_attributes().items():
            full = ent.name() + "." + key
            p[full] = type(p[full])(value.item())
        for key, value in ent.get_randomized_vec3_attributes().items():
            full = ent.name() + "." + key
            p[full] = type(p[full])(value.tolist())

    def _write_pose(self, ent) -> None:
        self._mitsuba_params[ent.name() + ".to_world"] = mi.Transform4f(ent._world_host().tolist())

    def update_camera(self) -> None:
        if not self._camera.randomizable():
            return
        self._write_pose(self._camera)
        self._write_attributes(self._camera)

    def update_projector(self) -> None:
        if not self._projector.randomizable():
            return
        self._write_pose(self._projector)
        self._write_attributes(self._projector)

    def update_lights(self) -> None:
        for light in self._lights:
            if not light.randomizable():
                continue
            if light.name() + ".to_world" in self._mitsuba_params.keys():
                self._write_pose(light)
            self._write_attributes(light)

    def update_materials(self) -> None:
        for mat in self._materials:
            if mat.randomizable():
                self._write_attributes(mat)

    # ------------------------------------------------------------------ randomisation
    def randomize_list(self, entity_list: List[entity.Transformable]) -> None:
        for root in [e for e in entity_list if e.parent() is None]:
            root.randomize()
            child = root.child()
            while child is not None:
                child.randomize()
                child = child.child()

    def randomize(self) -> None:
        self.randomize_list(self._meshes)
        self.randomize_list(self._lights)
        self.randomize_list(self._materials)
        if self._camera is not None:
            self._camera.randomize()
        if self._projector is not None:
            self._projector.randomize()
        self.update_meshes()
        if self._camera is not None:
            self.update_camera()
        if self._projector is not None:
            self.update_projector()
        self.update_lights()
        self.update_materials()
        self._mitsuba_params.update()
