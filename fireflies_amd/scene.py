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
import ctypes as C
from typing import List

import torch

from . import emitter, entity, material, mi, ops, sampling
from . import scenes as _scenes


def _seed_generators(seed: int) -> None:
    """what torch.manual_seed(seed) does to the generators that exist here — the CPU default generator and the CUDA
    default generators — without its detour through the lazy-initialisation hooks of the other backends
    (torch.xpu's `_lazy_call` formats a Python traceback on every call: ~50 us per scene sample)"""
    torch.default_generator.manual_seed(seed)
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        for g in torch.cuda.default_generators:
            g.manual_seed(seed)
    elif torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)  # queued until the device is initialised, as torch.manual_seed would


class _NativePlan:
    """the compiled form of one sampler configuration for ffx_scene_randomize_h (Scene._native_plan)"""

    @staticmethod
    def build(scene):
        from . import _abi
        from ._lib import api

        ents = list(scene._draw_order())
        row_of = {id(e): i for i, e in enumerate(ents)}
        draws, samplers, rows, posed, attrs = [], [], [], [], []
        dev = None

        def add(smp):
            nonlocal dev
            if type(smp) not in (sampling.UniformSampler, sampling.UniformScalarToVec3Sampler) or not smp._train:
                return None
            a = smp._min_range
            if not a.is_cuda or a.numel() < 1 or a.numel() > 4 or a.size() != smp._max_range.size() or smp._max_range.device != a.device:
                return None
            if dev is None:
                dev = a.device
            elif a.device != dev:
                return None
            lo, hi = smp._host_bounds()
            d = _abi.RandDraw()
            d.n = int(a.numel())
            for j in range(d.n):
                d.lo[j], d.hi[j] = float(lo[j]), float(hi[j])
            draws.append(d)
            samplers.append(smp)
            return len(draws) - 1

        for i, e in enumerate(ents):
            cls = type(e)
            if cls not in (entity.Transformable, entity.Mesh, emitter.Light, material.Material):
                return None  # a user subclass may override _draw / _compose
            kind = 0 if cls is material.Material else (2 if cls is entity.Mesh else 1)
            r = _abi.RandEntity()
            r.kind, r.parent, r.draw_t, r.draw_r, r.draw_s = kind, -1, -1, -1, -1
            par = e.parent()
            if par is not None:
                if id(par) not in row_of or row_of[id(par)] >= i:
                    return None
                r.parent = row_of[id(par)]
            c = e._centroid_mat
            r.centroid[0], r.centroid[1], r.centroid[2] = float(c[0, 3]), float(c[1, 3]), float(c[2, 3])
            src = e._world
            # (a Material draws its attributes whether it is marked randomisable or not — material.Material._draw has no such test,
            # the draws are part of the stream — and Scene.update_materials then writes them back only if it is)
            if e.randomizable() or kind == 0:
                if kind != 0:
                    dt, dr = add(e._translation_sampler), add(e._rotation_sampler)
                    if dt is None or dr is None:
                        return None
                    r.draw_t, r.draw_r = dt, dr
                    if kind == 2:
                        ds = add(e._scale_sampler)
                        if ds is None:
                            return None
                        r.draw_s = ds
                    posed.append((i, e, dt, dr))
                if kind != 2:  # (a Mesh does not sample its attributes: mesh.py:141-150)
                    fl, v3 = [], []
                    for key, smp in e._float_attributes.items():
                        d = add(smp)
                        if d is None or draws[d].n != 1:
                            return None
                        fl.append((key, d))
                    for key, smp in e._vec3_attributes.items():
                        d = add(smp)
                        if d is None:
                            return None
                        rep = 3 if type(smp) is sampling.UniformScalarToVec3Sampler else 1
                        if (rep == 1 and draws[d].n != 3) or (rep == 3 and draws[d].n != 1):
                            return None
                        v3.append((key, d, rep))
                    attrs.append((e, fl, v3))
            else:
                src = e._randomized_world  # (its local matrix as it stands: what _world_host() multiplies)
            w = src.detach().to("cpu", torch.float32).reshape(-1).tolist()
            for j in range(16):
                r.world[j] = w[j]
            rows.append(r)
        if dev is None:
            return None  # nothing is drawn from the device generator: the Python path has nothing to win or lose
        plan = _NativePlan()
        plan.n_draws, plan.n_ents = len(draws), len(rows)
        plan.draws = (_abi.RandDraw * max(len(draws), 1))(*draws)
        plan.ents = (_abi.RandEntity * max(len(rows), 1))(*rows)
        plan.posed, plan.attrs, plan.samplers = posed, attrs, samplers
        plan.mesh_rows = [row_of[id(m)] for m in scene._meshes]
        plan.row_of = row_of
        plan.gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        plan.seed1, plan.off1 = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
        fn = api().lib.ffx_scene_randomize_h
        plan.fn = fn
        plan.version_key = plan.versions()
        return plan

    def versions(self):
        """in-place writes into a sampler's bound tensors (through a tensor get_min() / get_max() handed out earlier) do not move the
        configuration counter: the tensors' own version counters do"""
        t = 0
        for s_ in self.samplers:
            t += s_._min_range._version + s_._max_range._version
        return t


class StaleDrawError(RuntimeError):
    """scene samples drawn ahead of time (Scene.randomize_batch(lazy=True)) no longer match the sampler configuration:
    the caller draws again.  Deliberately NOT raised for anything else — a failed native call or a HIP error inside the
    look-ahead must surface, not be retried."""


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
        self._native_chain = None  # set while _apply_native pushes a native call's tables (see _write_pose)
        self._pushed_natively = False  # set while _materialise tells the parameter map what the device already has
        self._lazy = None  # the natively pushed scene sample whose values the entities and the parameter map have not been told yet (_materialise)
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
            pick = mesh._pending_pick if hasattr(mesh, "_pending_pick") else mesh.sample_animation_index()
            world = mesh._world_host()
            uncentre = torch.eye(4)
            uncentre[0:3, 3] = -mesh._centroid_mat[0:3, 3]
            if pick is None:
                p.set_mesh_pose(mesh.name(), torch.from_numpy(entity.base.mm4(world, uncentre)), frame=0)
            elif pick[0] == "func":
                p.set_mesh_pose(mesh.name(), world, vertices=mesh._animation_func(mesh._vertices, pick[2]))
            else:
                pool = getattr(mesh, "_pool_frames", None)
                if pool is not None:  # frames already resident in the device pool, stored like the rest pose
                    p.set_mesh_pose(mesh.name(), torch.from_numpy(entity.base.mm4(world, uncentre)), frame=pool[pick[1]][0] + pick[2])
                else:
                    stack = mesh._anim_data_train if pick[1] == "train" else mesh._anim_data_eval
                    p.set_mesh_pose(mesh.name(), world, vertices=stack[pick[2]])

    def _write_attributes(self, ent) -> None:
        p = self._mitsuba_params
        # host copies made by the entity's _compose (one transfer per randomisation, not one sync per attribute)
        # (behind a native push — _apply_native — the values are only recorded: the device already has them)
        put = p._d.__setitem__ if self._pushed_natively else p.__setitem__
        for key, value in ent._host_float_attributes.items():
            full = ent.name() + "." + key
            put(full, type(p[full])(value))
        for key, value in ent._host_vec3_attributes.items():
            full = ent.name() + "." + key
            put(full, type(p[full])(value))

    def _write_pose(self, ent) -> None:
        nc = self._native_chain  # (inside _apply_native: the entity's world matrix is a row of the native call's chain table — _world_host()'s product, bit for bit)
        if nc is not None:
            i = nc[1].get(id(ent))
            if i is not None and hasattr(mi.Transform4f, "_from_rows"):
                if self._pushed_natively:
                    self._mitsuba_params._d[ent.name() + ".to_world"] = mi.Transform4f._from_rows(nc[0][i])
                else:
                    self._mitsuba_params[ent.name() + ".to_world"] = mi.Transform4f._from_rows(nc[0][i])
                return
        if self._pushed_natively:  # (told, not assigned: the device has it)
            self._mitsuba_params._d[ent.name() + ".to_world"] = mi.Transform4f(ent._world_host().tolist())
            return
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
        """reference API (scene.py:344-358): parents first, then the child chain — one entity at a time"""
        for ent in self._chain_order(entity_list):
            ent.randomize()

    @staticmethod
    def _chain_order(entity_list):
        for root in [e for e in entity_list if e.parent() is None]:
            yield root
            child = root.child()
            while child is not None:
                yield child
                child = child.child()

    def _draw_order(self):
        """every entity in the order Scene.randomize visits it (scene.py:360-371)"""
        yield from self._chain_order(self._meshes)
        yield from self._chain_order(self._lights)
        yield from self._chain_order(self._materials)
        if self._camera is not None:
            yield self._camera
        if self._projector is not None:
            yield self._projector

    def _draw_stream(self):
        """side stream for the sampler draws of CUDA entities: the device-to-host transfer of the drawn values
        then waits for the draws only, not for the render still running on the caller's stream.  (The Philox
        offsets of torch's CUDA generator are tracked on the host: the values do not depend on the stream.)"""
        dev = torch.device(self._device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(dev, priority=-1)  # high priority: a handful of tiny kernels ahead of the render
        return self._side

    def _draw_all(self, batch):
        """phase 1 for one scene sample: all sampler calls, reference order (entities, then the meshes' animation
        draws, which the reference makes inside update_meshes, scene.py:243-251 -> mesh.py:183-198)"""
        ents = list(self._draw_order())
        tickets = [e._draw(batch) for e in ents]
        picks = [m.sample_animation_index() if m.randomizable() else None for m in self._meshes]
        return ents, tickets, picks

    def _apply(self, drawn, values) -> None:
        """phase 2 + push: host algebra, parameter writes, ONE device pass (params.update())"""
        ents, tickets, picks = drawn
        for e, t in zip(ents, tickets):
            e._compose(t, values)
        for m, pick in zip(self._meshes, picks):
            m._pending_pick = pick
        try:
            self.update_meshes()
        finally:
            for m in self._meshes:
                del m._pending_pick
        if self._camera is not None:
            self.update_camera()
        if self._projector is not None:
            self.update_projector()
        self.update_lights()
        self.update_materials()
        self._mitsuba_params.update()

    def _fetch(self, batch):
        side = self._draw_stream()
        if side is None:
            return batch.fetch()
        with torch.cuda.stream(side):
            return batch.fetch()

    # ---- pre-drawn randomisation (CUDA entities, train mode).
    # A tiny kernel on ANY stream is not scheduled while a render that fills the GPU is in flight (measured:
    # ten torch.rand launches on a high-priority side stream complete ~0.6 ms later, when the render ends), so
    # a randomize() that draws on the device and then waits for the values serialises host and GPU (870
    # instead of 1 370 renders/s).  Therefore randomize() issues the draws of the NEXT randomisation right
    # away — before the caller launches this step's render — and rewinds the generators to where they were.
    # The next call uses those values only if it finds torch's CUDA generator, Python's `random` and the
    # sampler configuration exactly as this call left them (then it fast-forwards the generators to the state
    # after the pre-drawn calls): the numbers, and what any other consumer of the generators sees, are those
    # of the plain sequential program.  Otherwise the pre-drawn values are dropped and the draws are made now.
    # FFX_PREDRAW=0 switches the mechanism off.  (Caveat: sampler bounds written in place through a tensor
    # obtained earlier from get_min()/get_max() are not seen by the configuration check.)
    def _predraw_enabled(self):
        import os

        return self._draw_stream() is not None and os.environ.get("FFX_PREDRAW", "1") != "0" and all(e._train for e in self._draw_order())

    def _rng_state(self):
        """every generator a sampler may draw from: torch's CUDA and CPU generators (samplers with CPU bounds, a
        GaussianSampler with a CPU mean), Python's `random` (AnimationSampler) and numpy's global stream (custom samplers)"""
        import random as _random

        import numpy as _np

        return torch.cuda.get_rng_state(torch.device(self._device)), _random.getstate(), torch.get_rng_state(), _np.random.get_state()

    def _set_rng_state(self, st) -> None:
        import random as _random

        import numpy as _np

        torch.cuda.set_rng_state(st[0], torch.device(self._device))
        _random.setstate(st[1])
        torch.set_rng_state(st[2])
        _np.random.set_state(st[3])

    @staticmethod
    def _same_rng_state(a, b) -> bool:
        na, nb = a[3], b[3]
        return (torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[2], b[2])
                and na[0] == nb[0] and bool((na[1] == nb[1]).all()) and tuple(na[2:]) == tuple(nb[2:]))

    def _predraw(self) -> None:
        from .sampling import base as sbase

        side = self._draw_stream()
        before = self._rng_state()
        batch = entity.DrawBatch()
        with torch.cuda.stream(side):
            drawn = self._draw_all(batch)
            pending = batch.start_fetch()
        after = self._rng_state()
        self._set_rng_state(before)
        self._pre = {"before": before, "after": after, "drawn": drawn, "pending": pending, "config": sbase.mutation_count()}

    def _take_predrawn(self):
        from .sampling import base as sbase

        pre, self._pre = getattr(self, "_pre", None), None
        if pre is None or pre["config"] != sbase.mutation_count():
            return None
        if not self._same_rng_state(self._rng_state(), pre["before"]):
            return None
        self._set_rng_state(pre["after"])
        return pre["drawn"], pre["pending"].finish()

    def _host_drawable(self) -> bool:
        """True if every draw of a randomisation can be evaluated on the host (sampling/torch_rng.py): train mode and
        only uniform samplers with small bounds.  Cached until the sampler configuration changes."""
        from .sampling import base as sbase
        from .sampling import torch_rng

        c = getattr(self, "_hd", None)
        if c is None or c[0] != sbase.mutation_count():
            ok = torch_rng.verified(self._device)  # (one-time check against torch.rand on this device)
            if ok:
                for e in self._draw_order():
                    if not e.randomizable():
                        continue
                    extra = [e._scale_sampler] if hasattr(e, "_scale_sampler") else []  # meshes draw translation, rotation, scale
                    for smp in list(e._all_samplers()) + extra:
                        if type(smp) not in (sampling.UniformSampler, sampling.UniformScalarToVec3Sampler) or not smp._train or smp._min_range.numel() > 256:
                            ok = False
            c = self._hd = (sbase.mutation_count(), ok)
        return c[1]

    # ---- f1: the whole randomisation as ONE native call (include/ffx.h ffx_scene_randomize_h).
    # When every draw of a randomisation is a plain uniform draw from the device generator's stream (_host_drawable) and every entity
    # is one of the library's own classes, the order of the draws, their bounds and the way each entity's matrices follow from them
    # are fixed by the configuration: they are compiled once into two small tables (ffx_rand_draw / ffx_rand_entity), and a
    # randomisation is then a single host call that evaluates the Philox stream, the interval maps, the 4x4 algebra and the parent
    # chains for all the scene samples of a step — the Python per draw and per entity (0.17 ms per randomisation) is gone.  Anything
    # else — user samplers, Gaussian draws, eval mode, samplers on the CPU generator, subclasses that override _draw / _compose —
    # keeps the Python path below (per scene; FFX_NATIVE_RANDOMIZE=0 forces it).  Same values bit for bit: the native arithmetic is
    # the mirror's (tests/test_api_cpu.py), animation picks stay Python `random` draws made after the entity draws, as before.
    def _native_plan(self):
        import os

        from .sampling import base as sbase

        c = getattr(self, "_nplan", None)
        if c is not None and c[0] == sbase.mutation_count():
            plan = c[1]
            if plan is None or plan.versions() == plan.version_key:
                return plan
        plan = None
        if os.environ.get("FFX_NATIVE_RANDOMIZE", "1") != "0" and self._host_drawable():
            plan = _NativePlan.build(self)
        self._nplan = (sbase.mutation_count(), plan)
        return plan

    def _randomize_native(self, plan, seeds=None):
        """seeds None: one sample from the generator as it stands (advanced by the draws); a list: one sample per seed, each from offset 0
        of its seed (what `torch.manual_seed(s); randomize()` draws) — the generator is left seeded with the last one, advanced past its draws"""
        import numpy as np

        gen = plan.gen
        S = 1 if seeds is None else len(seeds)
        nd, ne = plan.n_draws, plan.n_ents
        if seeds is None:
            off = gen.get_offset()
            if off & 3:
                return None
            sa, oa = plan.seed1, plan.off1
            sa[0], oa[0] = gen.initial_seed(), off
        else:
            sa, oa = (C.c_uint64 * S)(*[int(s_) for s_ in seeds]), (C.c_uint64 * S)()
        vals = np.empty((S, nd, 4), np.float32)
        mats = np.empty((3, S, ne, 16), np.float32)  # local, chain, chain x un-centring
        rc = plan.fn(S, sa, oa, plan.draws, nd, plan.ents, ne, vals.ctypes.data, mats[0].ctypes.data, mats[1].ctypes.data, mats[2].ctypes.data)
        if rc != 0:
            return None
        if seeds is None:
            gen.set_offset(off + 4 * nd)
        return vals, mats

    def _apply_native(self, plan, vals, mats, k, picks) -> None:
        """phase 2 + push for sample k of a native call: entity state from the tables, parameter writes, ONE device pass"""
        if self._push_native(plan, vals, mats, k, picks):
            return
        self._materialise()  # (a pending sample first: this one then writes over it key by key)
        self._entity_state(plan, vals, mats, k)
        p = self._mitsuba_params
        if hasattr(p, "set_mesh_pose_np"):  # our parameter object: poses straight from the tables
            chain, unc = mats[1][k], mats[2][k]
            for m, pick, i in zip(self._meshes, picks, plan.mesh_rows):
                if not m.randomizable():
                    continue
                if pick is None:
                    p.set_mesh_pose_np(m.name(), unc[i], 0, None)
                elif pick[0] == "func":
                    p.set_mesh_pose_np(m.name(), chain[i], None, m._animation_func(m._vertices, pick[2]))
                else:
                    pool = getattr(m, "_pool_frames", None)
                    if pool is not None:
                        p.set_mesh_pose_np(m.name(), unc[i], pool[pick[1]][0] + pick[2], None)
                    else:
                        stack = m._anim_data_train if pick[1] == "train" else m._anim_data_eval
                        p.set_mesh_pose_np(m.name(), chain[i], None, stack[pick[2]])
        else:
            for m, pick in zip(self._meshes, picks):
                m._pending_pick = pick
            try:
                self.update_meshes()
            finally:
                for m in self._meshes:
                    del m._pending_pick
        self._native_chain = (mats[1][k], plan.row_of)
        try:
            if self._camera is not None:
                self.update_camera()
            if self._projector is not None:
                self.update_projector()
            self.update_lights()
            self.update_materials()
        finally:
            self._native_chain = None
        p.update()

    # ---- the native params.update() (include/ffx.h ffx_scene_step_h; mi.Scene.compile_step / step_native): the sample's tables go to the
    # device in ONE call — description, transform table, re-fit and pre-pass launches — and the parameter map is only told the values
    # afterwards (a script may read them back; nothing is pushed twice).  The reference writes key after key and lets Mitsuba's
    # params.update() find out what changed (fireflies/scene.py:243-342,384).  Compiled from what the Python path below was SEEN to do
    # with each key, hence available from the second sample of a configuration on; FFX_NATIVE_UPDATE=0 keeps the Python path.
    def _step_plan(self, plan):
        import os

        sp = getattr(plan, "step", None)
        if sp is not None or getattr(plan, "step_tries", 0) >= 4:
            return sp
        p = self._mitsuba_params
        ms = getattr(p, "_scene", None)
        if ms is None or not hasattr(ms, "compile_step") or os.environ.get("FFX_NATIVE_UPDATE", "1") == "0" or not getattr(self, "native_update", True):
            plan.step_tries = 4
            return None
        plan.step_tries = getattr(plan, "step_tries", 0) + 1
        keys = p.keys()
        pushed = [e for e in ([self._camera, self._projector] + list(self._lights)) if e is not None and e.randomizable()]
        poses = [(plan.row_of[id(e)], e.name() + ".to_world") for e in pushed if id(e) in plan.row_of and e.name() + ".to_world" in keys]
        writers = {id(e) for e in pushed + [m for m in self._materials if m.randomizable()]}
        values = []
        for e, fl, v3 in plan.attrs:
            if id(e) not in writers:
                continue
            values += [(d, int(plan.draws[d].n), 1, e.name() + "." + key) for key, d in fl]
            values += [(d, int(plan.draws[d].n), rep, e.name() + "." + key) for key, d, rep in v3]
        if any(full not in keys for _, _, _, full in values):
            plan.step_tries = 4  # (the Python path raises KeyError for it: let it)
            return None
        meshes = [(i, m.name()) for m, i in zip(self._meshes, plan.mesh_rows) if m.randomizable()]
        plan.step = ms.compile_step(poses, values, meshes, plan.n_draws, plan.n_ents)
        if plan.step is not None:
            for e in self._draw_order():
                e.__dict__["_lazy_owner"] = self  # (entity.base._Lazy: whom to ask for a pending sample's values)
        return plan.step

    def _push_native(self, plan, vals, mats, k, picks) -> bool:
        sp = self._step_plan(plan)
        if sp is None:
            return False
        frames = []
        for m, pick in zip(self._meshes, picks):
            if not m.randomizable():
                continue
            if pick is None:
                frames.append(0)
                continue
            pool = getattr(m, "_pool_frames", None)
            if pick[0] == "func" or pool is None:
                fb = self._mitsuba_params._scene.update_fallbacks
                fb["caller-supplied vertices"] = fb.get("caller-supplied vertices", 0) + 1
                return False  # (they travel through the vertex pool: the Python path)
            frames.append(pool[pick[1]][0] + pick[2])
        p = self._mitsuba_params
        lz = self._lazy
        if lz is not None and lz[0] is not plan:
            self._materialise()  # (another configuration's sample: its entities need not be this one's)
        if p._pending is not None and lz is None:
            p._pending()  # (a sample pushed through ANOTHER Scene object over the same parameter map: told to the map before this one replaces it)
        if not p._scene.step_native(sp, vals[k], mats[1][k], mats[2][k], frames):
            return False
        # what the sample means for the parameter map and the entities is worked out when somebody looks (_materialise): the device has it all
        self._lazy = (plan, vals, mats, k, sp)
        p._pending = self._materialise
        return True

    def _materialise(self) -> None:
        """the pending natively pushed sample -> entity state and parameter map (not dirty: nothing is pushed twice)"""
        lz = self._lazy
        if lz is None:
            return
        plan, vals, mats, k, sp = lz
        p = self._mitsuba_params
        self._lazy = None
        p._pending = None
        v = self._entity_state(plan, vals, mats, k)
        self._native_chain = (mats[1][k], plan.row_of)
        self._pushed_natively = True
        try:
            if self._camera is not None:
                self.update_camera()
            if self._projector is not None:
                self.update_projector()
            self.update_lights()
            self.update_materials()
        finally:
            self._native_chain = None
            self._pushed_natively = False
        for eta_key, drow in sp.eta:
            p._d[eta_key] = mi.Float(_scenes.specular_to_eta(v[drow][0]))

    @staticmethod
    def _entity_state(plan, vals, mats, k):
        v = vals[k].tolist()
        loc = mats[0][k]
        for i, e, dt, dr in plan.posed:
            e._randomized_world = torch.from_numpy(loc[i].reshape(4, 4))
            e._last_draw = (v[dt][:3], v[dr][:3])
        for e, fl, v3 in plan.attrs:
            e._host_float_attributes = {key: v[d][0] for key, d in fl}
            e._host_vec3_attributes = {key: (v[d][:3] if rep == 1 else [v[d][0]] * 3) for key, d, rep in v3}
            e._randomized_float_attributes = e._randomized_vec3_attributes = None
        return v

    def randomize(self) -> None:
        plan = self._native_plan()
        if plan is not None:
            out = self._randomize_native(plan)
            if out is not None:
                self._pre = None
                picks = [m.sample_animation_index() if m.randomizable() else None for m in self._meshes]
                self._apply_native(plan, out[0], out[1], 0, picks)
                return
        if self._host_drawable():
            # f1: no device work at all for the draws — values from the generator's Philox stream on the host,
            # 4x4 algebra on the host, matrices as kernel arguments of the one refit pass
            self._pre = None
            batch = entity.DrawBatch()
            drawn = self._draw_all(batch)
            self._apply(drawn, self._fetch(batch) if batch.needs_transfer() else batch.fetch())
            return
        side = self._draw_stream()
        got = self._take_predrawn() if side is not None else None
        if got is None:
            batch = entity.DrawBatch()
            if side is None:
                drawn = self._draw_all(batch)
            else:
                with torch.cuda.stream(side):
                    drawn = self._draw_all(batch)
            values = self._fetch(batch)
        else:
            drawn, values = got
        if side is not None:
            ops._stream_obj(side.device).wait_stream(side)  # drawn tensors may be used by animation functions
            if self._predraw_enabled():
                self._predraw()
        self._apply(drawn, values)

    def randomize_batch(self, seeds, lazy: bool = False):
        """The randomisations of several scene samples (BASELINE configs[3]: the samples of one optimisation step),
        each drawn under its own seed exactly as `torch.manual_seed(s); random.seed(s); randomize()` would, but
        with ONE device-to-host transfer for all of them.  Returns one callable per seed; calling it applies that
        sample (host algebra + parameter writes + params.update()), to be followed by the render of that sample.
        lazy=True returns a zero-argument function producing that list instead: the draws are enqueued now (put
        them in front of the renders they must not wait for), the transfer is awaited when it is called, and the
        generators are left as they were found (the caller is drawing ahead of time)."""
        import random as _random

        from .sampling import base as sbase

        plan = self._native_plan()
        if plan is not None and len(seeds) > 0:
            # one native call for the draws and matrices of all the samples; per sample the animation picks under its own `random` seed
            keep = None
            if lazy:
                keep = (torch.cuda.get_rng_state(torch.device(self._device)), torch.get_rng_state(), _random.getstate())
            out = self._randomize_native(plan, [int(s_) for s_ in seeds])
            if out is not None:
                all_picks = []
                for seed in seeds:
                    _random.seed(int(seed))
                    all_picks.append([m.sample_animation_index() if m.randomizable() else None for m in self._meshes])
                # the generators as `manual_seed(last); randomize()` leaves them
                _seed_generators(int(seeds[-1]))
                plan.gen.set_offset(4 * plan.n_draws)
                config = sbase.mutation_count()
                drawn_versions = plan.version_key  # (_native_plan has just checked it against the bound tensors)
                if keep is not None:
                    torch.cuda.set_rng_state(keep[0], torch.device(self._device))
                    torch.set_rng_state(keep[1])
                    _random.setstate(keep[2])

                def native_appliers():
                    # (an in-place edit of a sampler's bound tensor — through a handle get_min() / get_max() gave out earlier — does not move the
                    # configuration counter: the tensors' own version counters do, round-4 advisor)
                    if sbase.mutation_count() != config or plan.versions() != drawn_versions:
                        raise StaleDrawError("the sampler configuration changed after these scene samples were drawn")
                    return [(lambda k=k: self._apply_native(plan, out[0], out[1], k, all_picks[k])) for k in range(len(seeds))]

                return native_appliers if lazy else native_appliers()
        side = None if self._host_drawable() else self._draw_stream()  # host-evaluated draws: no device work to order
        keep = None
        if lazy:
            on_gpu = torch.device(self._device).type == "cuda" and torch.cuda.is_available()  # (host-evaluated draws advance that generator too)
            keep = (torch.cuda.get_rng_state(torch.device(self._device)) if on_gpu else None, torch.get_rng_state(), _random.getstate())
        batch = entity.DrawBatch()
        drawn = []
        for seed in seeds:
            _seed_generators(int(seed))
            _random.seed(int(seed))
            if side is None:
                drawn.append(self._draw_all(batch))
            else:
                with torch.cuda.stream(side):
                    drawn.append(self._draw_all(batch))
        if side is None:
            pending = batch.start_fetch()
        else:
            with torch.cuda.stream(side):
                pending = batch.start_fetch()
        config = sbase.mutation_count()
        if keep is not None:
            if keep[0] is not None:
                torch.cuda.set_rng_state(keep[0], torch.device(self._device))
            torch.set_rng_state(keep[1])
            _random.setstate(keep[2])

        def appliers():
            if sbase.mutation_count() != config:
                raise StaleDrawError("the sampler configuration changed after these scene samples were drawn")
            values = pending.finish()
            if side is not None:
                ops._stream_obj(side.device).wait_stream(side)
            return [(lambda d=d: self._apply(d, values)) for d in drawn]

        return appliers if lazy else appliers()
