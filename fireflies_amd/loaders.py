"""Scene-file readers (host side): a subset of the Mitsuba-3 XML format and Wavefront OBJ.

SURVEY §8(f) f4: the reference loads its scenes with `mi.load_file("…/vocalfold.xml")`
(examples/vocalfold_scene.py:20-22) and its animation frames with pywavefront
(fireflies/entity/mesh.py:167-181).  Neither Mitsuba nor pywavefront is a dependency here; this
module reads what those scene files contain so that `mi.load_file(path)` keeps working:

  <sensor type="perspective">   fov, near_clip, far_clip, <transform name="to_world">, <film> width/height
  <shape type="obj"|"ply">       filename (Wavefront OBJ; Stanford PLY ascii / binary), optional to_world (baked into the vertices),
                                 nested or referenced <bsdf>: diffuse reflectance, or principled with its scalar parameters (scenes.PRINCIPLED_DEFAULTS)
  <emitter type="spot">          intensity, cutoff_angle, beam_width, to_world
  <emitter type="projector">     irradiance texture (id "tex"), scale, fov, to_world
  <transform>                    <matrix value="16 floats">, <lookat origin target up>, <translate>, <scale>,
                                 <rotate x|y|z angle>  (applied in document order, like Mitsuba)
  <default name value>, $name substitution, <integer|float|string|rgb|boolean name value>
  <integrator>                   read for what it implies: direct illumination at the primary hit is what is rendered
                                 (path / prb / direct with max_depth 2); anything deeper is reported
  <film> <rfilter>, <sampler>    box and gaussian (the hdrfilm default) filters and the independent sampler are implemented; others are reported
  fov_axis                       x (Mitsuba's default), y, smaller, larger, diagonal: converted to the horizontal angle

Every node or property that is dropped is reported with a warning (once per kind and file); nothing here touches the GPU.
"""
import os
import re
import warnings
import xml.etree.ElementTree as ET

import numpy as np

from . import scenes


# ----------------------------------------------------------------------------- OBJ
def load_obj(path, with_info=False, split_seams=True):
    """-> (vertices [V,3] float32, triangles [F,3] int32).  Faces with more than three corners are
    fan-triangulated; `v/vt/vn` corner syntax and negative (relative) indices are handled.
    with_info: a third item {"has_normals": the file carries `vn` records, "uv": [V,2] texture coordinates or None} —
    Mitsuba shades a mesh with `vn` with interpolated vertex normals (re-derived from the positions after every vertex
    update: include/ffx.h ffx_smooth; the file's own normal VALUES are therefore not kept).  Texture coordinates are per
    VERTEX here as in Mitsuba's mesh: a vertex that faces use with different `vt` (a seam) is duplicated once per extra
    `vt` (appended after the file's vertices, so that vertex i < V keeps its index); v is flipped (v -> 1 - v: Mitsuba's
    `flip_tex_coords` default for OBJ)."""
    verts, tris, vts = [], [], []
    corner_vt = []  # per triangle corner: vt index or -1
    has_vn = False
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vn "):
                has_vn = True
            elif line.startswith("vt "):
                p = line.split()
                vts.append((float(p[1]), float(p[2]) if len(p) > 2 else 0.0))
            elif line.startswith("f "):
                idx, tix = [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                    if len(parts) > 1 and parts[1] != "":
                        k = int(parts[1])
                        tix.append(k - 1 if k > 0 else len(vts) + k)
                    else:
                        tix.append(-1)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
                    corner_vt.append((tix[0], tix[k], tix[k + 1]))
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    t = np.asarray(tris, np.int32).reshape(-1, 3)
    if t.size and (t.min() < 0 or t.max() >= v.shape[0]):
        raise ValueError(f"{path}: face index out of range")
    if not with_info:
        return v, t
    uv = None
    cv = np.asarray(corner_vt, np.int64).reshape(-1, 3)
    # split_seams=False: the caller has no use for texture coordinates (no textured base colour bound to the shape) — the file's vertex
    # list is kept as it is, so that OBJ animation frames and user `vertex_positions` (which carry the file's V vertices, as the reference
    # feeds them: fireflies/entity/mesh.py:167-181) still fit the mesh
    if split_seams and len(vts) and cv.size and (cv >= 0).all():
        if cv.max() >= len(vts):
            raise ValueError(f"{path}: texture-coordinate index out of range")
        vt = np.asarray(vts, np.float32).reshape(-1, 2)
        first = {}  # vertex -> its first vt; (vertex, other vt) -> duplicated vertex
        extra_v, extra_uv = [], []
        uv_of = np.full(v.shape[0], -1, np.int64)
        t = t.copy()
        for f_ in range(t.shape[0]):
            for c in range(3):
                vi, ti = int(t[f_, c]), int(cv[f_, c])
                if uv_of[vi] < 0:
                    uv_of[vi] = ti
                elif uv_of[vi] != ti:  # a seam: this corner needs its own copy of the vertex
                    key = (vi, ti)
                    if key not in first:
                        first[key] = v.shape[0] + len(extra_v)
                        extra_v.append(v[vi])
                        extra_uv.append(ti)
                    t[f_, c] = first[key]
        uv_idx = np.concatenate([np.where(uv_of >= 0, uv_of, 0), np.asarray(extra_uv, np.int64)]) if extra_uv else np.where(uv_of >= 0, uv_of, 0)
        if extra_v:
            v = np.concatenate([v, np.asarray(extra_v, np.float32)], 0)
        uv = vt[uv_idx].copy()
        uv[:, 1] = 1.0 - uv[:, 1]
    return v, t, {"has_normals": has_vn, "uv": uv}


# ----------------------------------------------------------------------------- PLY
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path, with_info=False):
    """Stanford PLY (ascii, binary_little_endian, binary_big_endian) -> (vertices [V,3] float32, triangles [F,3]
    int32).  Reads the `vertex` element's x/y/z and the `face` element's index list (vertex_indices /
    vertex_index), fan-triangulating polygons; other elements and properties are skipped.  `Scene` classifies
    parameter keys containing "ply" as meshes exactly like "mesh" (fireflies/scene.py:100), and Mitsuba scene
    files reference PLY shapes as <shape type="ply">."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append(("scalar", tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
        verts, tris = None, []
        if fmt == "ascii":
            tokens = iter(f.read().split())

            def scalar(t):
                v = next(tokens)
                return float(v) if _PLY_TYPES[t][0] == "f" else int(v)

        else:
            end = "<" if fmt == "binary_little_endian" else ">"

            def scalar(t):
                dt = np.dtype(end + _PLY_TYPES[t])
                return np.frombuffer(f.read(dt.itemsize), dt)[0].item()

        for el in elements:
            names = [p[-1] for p in el["props"]]
            all_scalar = all(p[0] == "scalar" for p in el["props"])
            if el["name"] == "vertex":
                if not {"x", "y", "z"} <= set(names):
                    raise ValueError(f"{path}: vertex element without x/y/z")
                if fmt != "ascii" and all_scalar:  # one structured read
                    dt = np.dtype([(p[2], end + _PLY_TYPES[p[1]]) for p in el["props"]])
                    raw = np.frombuffer(f.read(dt.itemsize * el["count"]), dt)
                    verts = np.stack([raw["x"], raw["y"], raw["z"]], 1).astype(np.float32)
                else:
                    rows = np.empty((el["count"], 3), np.float32)
                    for i in range(el["count"]):
                        rec = {}
                        for p in el["props"]:
                            if p[0] == "scalar":
                                rec[p[2]] = scalar(p[1])
                            else:
                                rec[p[3]] = [scalar(p[2]) for _ in range(int(scalar(p[1])))]
                        rows[i] = (rec["x"], rec["y"], rec["z"])
                    verts = rows
            else:
                for _ in range(el["count"]):
                    for p in el["props"]:
                        if p[0] == "scalar":
                            scalar(p[1])
                            continue
                        idx = [int(scalar(p[2])) for _ in range(int(scalar(p[1])))]
                        if el["name"] == "face" and p[3] in ("vertex_indices", "vertex_index"):
                            for k in range(1, len(idx) - 1):
                                tris.append((idx[0], idx[k], idx[k + 1]))
    if verts is None:
        raise ValueError(f"{path}: no vertex element")
    t = np.asarray(tris, np.int32).reshape(-1, 3)
    if t.size and (t.min() < 0 or t.max() >= verts.shape[0]):
        raise ValueError(f"{path}: face index out of range")
    if with_info:  # (vertex normals nx/ny/nz: shaded with interpolated normals, like an OBJ with `vn`)
        has_vn = any(el["name"] == "vertex" and {"nx", "ny", "nz"} <= {p[-1] for p in el["props"]} for el in elements)
        return verts, t, {"has_normals": has_vn, "uv": None}
    return verts, t


def save_ply(path, verts, tris, binary=True):
    verts, tris = np.asarray(verts, np.float32), np.asarray(tris, np.int32)
    head = (f"ply\nformat {'binary_little_endian' if binary else 'ascii'} 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\n"
            f"property float z\nelement face {len(tris)}\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if binary:
            f.write(verts.astype("<f4").tobytes())
            rec = np.empty(len(tris), np.dtype([("n", "u1"), ("i", "<i4", 3)]))
            rec["n"], rec["i"] = 3, tris
            f.write(rec.tobytes())
        else:
            for v in verts:
                f.write(f"{v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n".encode())
            for t in tris:
                f.write(f"3 {t[0]} {t[1]} {t[2]}\n".encode())


def load_mesh_file(path):
    """OBJ or PLY by extension"""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".ply":
        return load_ply(path)
    if ext == ".obj":
        return load_obj(path)
    raise NotImplementedError(f"{path}: only .obj and .ply meshes are read")


def save_obj(path, verts, tris):
    with open(path, "w") as f:
        for v in np.asarray(verts):
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for t in np.asarray(tris):
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


def load_obj_sequence(directory):
    """all *.obj of a directory in sorted order -> frames [T,V,3] (same topology assumed) and the
    triangles of the first file (fireflies/entity/mesh.py:167-181)."""
    files = sorted(f for f in os.listdir(directory) if f.endswith(".obj"))
    if not files:
        raise FileNotFoundError(f"no .obj files in {directory}")
    frames, tris = [], None
    for f in files:
        v, t = load_obj(os.path.join(directory, f))
        if tris is None:
            tris = t
        elif v.shape != frames[0].shape:
            raise ValueError(f"{f}: vertex count differs from the first frame")
        frames.append(v)
    return np.stack(frames), tris


# ----------------------------------------------------------------------------- XML helpers
def _floats(s):
    return [float(x) for x in re.split(r"[,\s]+", s.strip()) if x]


def _rot(axis, deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    m = np.eye(4)
    m[:3, :3] = np.eye(3) * c + s * K + (1 - c) * np.outer(a, a)
    return m


def _transform(node):
    m = np.eye(4)
    if node is None:
        return m.astype(np.float32)
    for op in node:
        if op.tag == "matrix":
            t = np.asarray(_floats(op.get("value")), np.float64).reshape(4, 4)
        elif op.tag == "lookat":
            t = scenes.look_at(_floats(op.get("origin")), _floats(op.get("target")), _floats(op.get("up", "0,1,0"))).astype(np.float64)
        elif op.tag == "translate":
            t = np.eye(4)
            t[:3, 3] = _floats(op.get("value")) if op.get("value") else [float(op.get(k, 0)) for k in "xyz"]
        elif op.tag == "scale":
            v = _floats(op.get("value")) if op.get("value") else [float(op.get(k, 1)) for k in "xyz"]
            v = v * 3 if len(v) == 1 else v
            t = np.diag([v[0], v[1], v[2], 1.0])
        elif op.tag == "rotate":
            axis = _floats(op.get("value")) if op.get("value") else [float(op.get(k, 0)) for k in "xyz"]
            t = _rot(axis, float(op.get("angle")))
        else:
            warnings.warn(f"transform op <{op.tag}> ignored")
            continue
        m = t @ m  # later operations are applied after earlier ones
    return m.astype(np.float32)


def _props(node):
    out = {}
    for c in node:
        if c.tag in ("float", "integer", "string", "boolean") and c.get("name"):
            v = c.get("value")
            out[c.get("name")] = float(v) if c.tag == "float" else int(v) if c.tag == "integer" else (v == "true") if c.tag == "boolean" else v
        elif c.tag in ("rgb", "spectrum") and c.get("name"):
            v = _floats(c.get("value"))
            out[c.get("name")] = tuple(v * 3 if len(v) == 1 else v[:3])
    return out


def _child(node, tag, name=None):
    for c in node:
        if c.tag == tag and (name is None or c.get("name") == name):
            return c
    return None


def _base_texture_of(bsdf, bsdfs, base_dir):
    """the [h,w,3] float32 texture of a bsdf's texture-valued base colour (`<texture type="bitmap" name="base_color">`), or None.
    The bitmap is read with PIL when it is there (sRGB -> linear like Mitsuba's default `raw` = false); a missing file or
    decoder gives a mid-grey 1x1 texture with a warning — the parameter `<mat>.brdf_0.base_color.data` exists either way, which
    is what the reference's loop assigns to (main.py:147-153)."""
    if bsdf is None:
        return None
    if bsdf.tag == "ref":
        return _base_texture_of(bsdfs.get(bsdf.get("id")), bsdfs, base_dir)
    inner = bsdf
    while inner.get("type") in ("twosided", "bumpmap", "normalmap", "mask") and _child(inner, "bsdf") is not None:
        inner = _child(inner, "bsdf")
    for c in inner:
        if c.tag == "texture" and c.get("name") in ("base_color", "reflectance"):
            if c.get("type") != "bitmap":
                warnings.warn(f"texture type {c.get('type')!r} for the base colour: a mid-grey 1x1 texture stands in for it")
                return np.full((1, 1, 3), 0.5, np.float32)
            fn = _props(c).get("filename")
            path = os.path.join(base_dir, fn) if fn else None
            try:
                from PIL import Image  # (not a dependency: used when present)

                img = np.asarray(Image.open(path).convert("RGB"), np.float32) / 255.0
                if not bool(_props(c).get("raw", False)):  # sRGB -> linear
                    img = np.where(img <= 0.04045, img / 12.92, ((img + 0.055) / 1.055) ** 2.4)
                return np.ascontiguousarray(img, np.float32)
            except Exception as e:  # noqa: BLE001 — any reason the bitmap cannot be had
                warnings.warn(f"base-colour bitmap {fn!r} could not be read ({type(e).__name__}): a mid-grey 1x1 texture stands in for it; "
                              "assign params['<mat>.brdf_0.base_color.data']")
                return np.full((1, 1, 3), 0.5, np.float32)
    return None


def _albedo_of(bsdf, bsdfs):
    """-> (base colour, material id, principled parameters or None for a diffuse material)"""
    if bsdf is None:
        return (0.5, 0.5, 0.5), "mat-Default", None
    if bsdf.tag == "ref":
        ref = bsdfs.get(bsdf.get("id"))
        if ref is None:
            raise ValueError(f"unknown bsdf reference {bsdf.get('id')}")
        col, _, params = _albedo_of(ref, bsdfs)
        return col, bsdf.get("id"), params
    inner = bsdf
    while inner.get("type") in ("twosided", "bumpmap", "normalmap", "mask") and _child(inner, "bsdf") is not None:
        inner = _child(inner, "bsdf")
    p = _props(inner)
    for c in inner:
        if c.tag == "texture" and c.get("name") and c.get("name") not in ("base_color", "reflectance"):
            warnings.warn(f"bsdf {bsdf.get('id') or inner.get('type')!r}: texture-valued parameter {c.get('name')!r} is not evaluated: its default is used "
                          "(only the base colour can be a texture)")
    col = p.get("reflectance", p.get("base_color", p.get("diffuse_reflectance", (0.5, 0.5, 0.5))))
    if isinstance(col, float):
        col = (col,) * 3
    params = None
    if inner.get("type") == "principled":  # the plugin's scalar parameters (textures are not read); `eta` only without `specular`
        params = {k: float(p[k]) for k in list(scenes.PRINCIPLED_DEFAULTS) + ["eta"] if isinstance(p.get(k), (int, float))}
    elif inner.get("type") != "diffuse":
        warnings.warn(f"bsdf type {inner.get('type')!r} is rendered as Lambert with its base colour (DESIGN.md §4.3)")
    return tuple(col), bsdf.get("id") or "mat-Default", params


_SENSOR_PROPS = {"fov", "fov_axis", "near_clip", "far_clip", "focus_distance", "principal_point_offset_x", "principal_point_offset_y"}
_FILM_PROPS = {"width", "height", "pixel_format", "component_format", "file_format", "sample_border", "compensate", "crop_offset_x", "crop_offset_y", "crop_width", "crop_height"}
_SHAPE_PROPS = {"filename", "face_normals", "flip_tex_coords", "flip_normals"}
_SPOT_PROPS = {"intensity", "cutoff_angle", "beam_width"}
_PROJ_PROPS = {"fov", "scale", "fov_axis"}


def _fov_x(fov, axis, width, height):
    """the horizontal field of view (degrees) of a perspective sensor whose `fov` is measured along `fov_axis`
    (Mitsuba: x [default], y, diagonal, smaller, larger)"""
    if axis in (None, "x"):
        return float(fov)
    w, h = float(width), float(height)
    if axis == "smaller":
        axis = "x" if w <= h else "y"
    elif axis == "larger":
        axis = "x" if w >= h else "y"
    if axis == "x":
        return float(fov)
    t = np.tan(np.deg2rad(float(fov)) * 0.5)
    if axis == "y":
        tx = t * w / h
    elif axis == "diagonal":
        tx = t * w / np.hypot(w, h)
    else:
        raise ValueError(f"unknown fov_axis {axis!r}")
    return float(np.rad2deg(2.0 * np.arctan(tx)))


class _Dropped:
    """collects what the reader does not honour and reports each kind once"""

    def __init__(self, path):
        self.path, self.seen = path, set()

    def __call__(self, kind, msg):
        if kind not in self.seen:
            self.seen.add(kind)
            warnings.warn(f"{os.path.basename(self.path)}: {msg}", stacklevel=4)

    def props(self, node, known, what):
        for c in node:
            if c.tag in ("float", "integer", "string", "boolean", "rgb", "spectrum", "point", "vector") and c.get("name") and c.get("name") not in known:
                self((what, c.get("name")), f"{what} property {c.get('name')!r} is ignored")


def _check_integrator(node, dropped, notes):
    """direct illumination at the primary hit from the two delta emitters is what the kernels evaluate: Mitsuba's path / prb
    with max_depth = 2, or `direct`.  Mitsuba's defaults (max_depth -1 = unbounded) and anything deeper would add indirect
    light that is not rendered here: say so."""
    t = node.get("type")
    p = _props(node)
    notes["integrator"] = {"type": t, **{k: v for k, v in p.items() if isinstance(v, (int, float, str, bool))}}
    if t in ("path", "prb", "prb_basic", "volpath", "prbvolpath", "prb_reparam", "direct_reparam", "direct_projective", "prb_projective"):
        md = int(p.get("max_depth", -1))
        if md < 0 or md > 2:
            dropped("integrator.max_depth", f"<integrator type={t!r}> with max_depth {md if md >= 0 else '-1 (unbounded)'}: only direct illumination at the "
                                            "primary hit is rendered (= max_depth 2); indirect bounces are not")
        if t.endswith("reparam") or t.endswith("projective"):
            dropped("integrator.reparam", f"<integrator type={t!r}>: visibility-discontinuity gradients are not computed (gradients flow through the projector texture only)")
    elif t == "direct":
        pass
    elif t in ("aov", "moment", "stokes"):
        dropped("integrator.type", f"<integrator type={t!r}> wrapper ignored: only the radiance image is rendered")
    else:
        dropped("integrator.type", f"<integrator type={t!r}> is not implemented: direct illumination at the primary hit is rendered instead")


def load_mitsuba_xml(path):
    """-> scenes.SceneData.  Sensor 0 is the camera; a second perspective sensor, if present, is the
    projector proxy whose film size is the projector texture size (examples/vocalfold_scene.py:24-38).
    Whatever the file asks for that is not honoured is reported (warnings.warn), never dropped silently."""
    with open(path, "r") as f:
        text = f.read()
    defaults = dict(re.findall(r'<default\s+name="([^"]+)"\s+value="([^"]*)"', text))
    for k, v in defaults.items():
        text = text.replace("$" + k, v)
    root = ET.fromstring(text)
    base = os.path.dirname(os.path.abspath(path))
    bsdfs = {b.get("id"): b for b in root.iter("bsdf") if b.get("id")}
    sensors, meshes, spot, projector, proj_scale = [], [], None, None, 1.0
    dropped = _Dropped(path)
    notes = {"source": path}
    have_integrator = False
    for node in root:
        if node.tag == "sensor":
            if node.get("type") != "perspective":
                dropped(("sensor", node.get("type")), f"sensor type {node.get('type')!r} ignored")
                continue
            p = _props(node)
            dropped.props(node, _SENSOR_PROPS, "sensor")
            film = _child(node, "film")
            fp = _props(film) if film is not None else {}
            width, height = int(fp.get("width", 768)), int(fp.get("height", 576))
            if film is not None:
                dropped.props(film, _FILM_PROPS, "film")
                if film.get("type") not in (None, "hdrfilm"):
                    dropped(("film", film.get("type")), f"film type {film.get('type')!r}: an hdrfilm-like float RGB image is produced")
                if any(k in fp for k in ("crop_offset_x", "crop_offset_y", "crop_width", "crop_height")):
                    dropped("film.crop", "film crop window ignored: the full film is rendered")
                rf = _child(film, "rfilter")
                rtype = rf.get("type") if rf is not None else "gaussian"  # Mitsuba's default film filter
                if not sensors:  # (the camera's film; a projector proxy's film is only a texture size)
                    if rtype == "gaussian":  # implemented (ffx_render_fwd_filtered): what mi.Scene then renders with
                        stddev = float(_props(rf).get("stddev", 0.5)) if rf is not None else 0.5
                        if stddev > 0.5:
                            dropped("rfilter.stddev", f"gaussian reconstruction filter with stddev {stddev}: its radius does not fit the 5x5-pixel window; 0.5 is used")
                            stddev = 0.5
                        notes.setdefault("rfilter_stddev", stddev)
                    elif rtype != "box":
                        dropped("rfilter", f"reconstruction filter {rtype!r}: box and gaussian are implemented; the BOX filter is used — every sample counts "
                                           "for its own pixel only (equal to Mitsuba's image in the mean)")
                        rtype = "box"
                notes.setdefault("rfilter", rtype)
            for extra in node:
                if extra.tag == "sampler":
                    if extra.get("type") not in ("independent",):
                        dropped("sampler", f"sampler type {extra.get('type')!r}: the independent (counter-based, per-sample hash) sampler is used")
                    if "sample_count" in _props(extra):
                        notes.setdefault("sample_count", int(_props(extra)["sample_count"]))
                elif extra.tag not in ("film", "transform", "float", "integer", "string", "boolean", "ref"):
                    dropped(("sensor-child", extra.tag), f"<{extra.tag}> inside <sensor> ignored")
            if "principal_point_offset_x" in p or "principal_point_offset_y" in p:
                if float(p.get("principal_point_offset_x", 0.0)) != 0.0 or float(p.get("principal_point_offset_y", 0.0)) != 0.0:
                    dropped("principal_point", "principal_point_offset_x/y ignored: the principal point is the film centre")
            name = node.get("id") or ("PerspectiveCamera" if not sensors else f"PerspectiveCamera_{len(sensors)}")
            sensors.append(scenes.SensorData(name, _transform(_child(node, "transform", "to_world")), _fov_x(p.get("fov", 45.0), p.get("fov_axis"), width, height),
                                             float(p.get("near_clip", 0.01)), float(p.get("far_clip", 1e4)), width, height))
        elif node.tag == "integrator":
            have_integrator = True
            _check_integrator(node, dropped, notes)
        elif node.tag in ("bsdf", "default"):
            pass  # top-level BSDF definitions are resolved through <ref>; <default> was substituted above
        elif node.tag == "texture":
            dropped("texture", "top-level <texture> ignored: textures as BSDF parameter values are set through params['<mat>.brdf_0.base_color.data'] (mi.py), "
                               "bitmap files are not read")
        elif node.tag == "shape":
            if node.get("type") not in ("obj", "ply"):
                raise NotImplementedError(f"shape type {node.get('type')!r}: only OBJ and PLY meshes are read")
            p = _props(node)
            dropped.props(node, _SHAPE_PROPS, "shape")
            for extra in node:
                if extra.tag == "emitter":
                    dropped("area-emitter", "area emitter on a shape ignored: only the delta emitters (projector, spot) illuminate the scene")
                elif extra.tag in ("medium", "sensor"):
                    dropped(("shape-child", extra.tag), f"<{extra.tag}> inside <shape> ignored")
            bnode = _child(node, "bsdf") if _child(node, "bsdf") is not None else _child(node, "ref")
            btex = _base_texture_of(bnode, bsdfs, base)
            if node.get("type") == "ply":
                v, t, finfo = load_ply(os.path.join(base, p["filename"]), with_info=True)
            else:  # seam vertices are only duplicated for a shape that actually gets a textured base colour
                v, t, finfo = load_obj(os.path.join(base, p["filename"]), with_info=True, split_seams=btex is not None)
            # Mitsuba: a mesh with vertex normals is shaded in the interpolated frame unless face_normals is set
            smooth = bool(finfo["has_normals"]) and not bool(p.get("face_normals", False))
            M = _transform(_child(node, "transform", "to_world"))
            v = (v @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
            alb, mat, bsdf = _albedo_of(_child(node, "bsdf") if _child(node, "bsdf") is not None else _child(node, "ref"), bsdfs)
            uv = finfo.get("uv")
            if btex is not None and uv is None:
                dropped(("tex-without-uv", node.get("id")), f"shape {node.get('id')!r}: a textured base colour but no texture coordinates in the mesh file: the constant base colour is used")
                btex = None
            meshes.append(scenes.MeshData(node.get("id") or f"mesh-{len(meshes)}", v[None], t, alb, mat, bsdf, smooth=smooth, uv=uv, base_tex=btex))
        elif node.tag == "emitter":
            p = _props(node)
            tw = _transform(_child(node, "transform", "to_world"))
            if node.get("type") == "spot":
                dropped.props(node, _SPOT_PROPS, "spot emitter")
                if _child(node, "texture") is not None:
                    dropped("spot.texture", "spot emitter texture ignored: a plain spot light is rendered")
                inten = p.get("intensity", (1.0, 1.0, 1.0))
                inten = (inten,) * 3 if isinstance(inten, float) else inten
                cutoff = float(p.get("cutoff_angle", 20.0))
                spot = scenes.SpotData(node.get("id") or "emit-Spot", tw, tuple(inten), cutoff, float(p.get("beam_width", cutoff * 0.75)))
            elif node.get("type") == "projector":
                dropped.props(node, _PROJ_PROPS, "projector emitter")
                projector = (tw, float(p.get("fov", 45.0)), p.get("fov_axis"))
                proj_scale = float(p.get("scale", 1.0))
            else:
                dropped(("emitter", node.get("type")), f"emitter type {node.get('type')!r} ignored: only `projector` and `spot` emitters illuminate the scene")
        else:
            dropped(("top", node.tag), f"top-level <{node.tag}> ignored")
    if not have_integrator:
        dropped("integrator.default", "no <integrator>: Mitsuba would run `path` with unbounded depth; only direct illumination at the primary hit is rendered (= max_depth 2)")
    if not sensors:
        raise ValueError(f"{path}: no perspective sensor")
    if not meshes:
        raise ValueError(f"{path}: no shapes")
    proj = None
    if projector is not None:
        proxy = sensors[1] if len(sensors) > 1 else None
        w, h = (proxy.width, proxy.height) if proxy is not None else (500, 500)
        proj = scenes.SensorData(proxy.name if proxy is not None else "PerspectiveCamera_1", projector[0],
                                 proxy.fov_x if proxy is not None else _fov_x(projector[1], projector[2], w, h),
                                 proxy.near if proxy is not None else 0.01, proxy.far if proxy is not None else 1e4, w, h)
    for m in meshes:
        if m.bsdf is not None and float(m.bsdf.get("spec_trans", 0.0)) > 0.0:
            dropped("spec_trans", f"material {m.material!r}: spec_trans > 0 — the principled BSDF's transmission lobe is not evaluated; spec_trans only scales the "
                                  "diffuse lobe (DESIGN.md 4.3)")
    return scenes.SceneData(meshes, sensors[0], proj, spot, proj_scale, notes=notes)
