"""Scene-file readers: OBJ and the Mitsuba-XML subset.  Host side only."""
import numpy as np
import pytest

from fireflies_amd import loaders, scenes

XML = """<scene version="3.0.0">
  <default name="spp" value="64"/>
  <default name="res" value="96"/>
  <bsdf type="twosided" id="mat-Mucosa"><bsdf type="principled"><rgb name="base_color" value="0.8, 0.3, 0.35"/><float name="roughness" value="0.35"/><float name="specular" value="0.6"/><float name="clearcoat" value="0.25"/></bsdf></bsdf>
  <sensor type="perspective" id="PerspectiveCamera">
    <float name="fov" value="60"/><float name="near_clip" value="0.01"/><float name="far_clip" value="100"/>
    <transform name="to_world"><lookat origin="0,0,1.5" target="0,0,5" up="0,1,0"/></transform>
    <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="$res"/></film>
  </sensor>
  <sensor type="perspective" id="PerspectiveCamera_1">
    <float name="fov" value="30"/>
    <transform name="to_world"><lookat origin="0.25,0,1.5" target="0,0,5" up="0,1,0"/></transform>
    <film type="hdrfilm"><integer name="width" value="128"/><integer name="height" value="128"/></film>
  </sensor>
  <shape type="obj" id="mesh-Wall">
    <string name="filename" value="wall.obj"/>
    <transform name="to_world"><scale value="2"/><translate z="6"/></transform>
    <ref id="mat-Mucosa"/>
  </shape>
  <shape type="obj" id="mesh-Quad"><string name="filename" value="quad.obj"/>
    <bsdf type="diffuse"><rgb name="reflectance" value="0.2"/></bsdf></shape>
  <emitter type="spot" id="emit-Spot"><rgb name="intensity" value="8"/><float name="cutoff_angle" value="40"/>
    <transform name="to_world"><lookat origin="0,0.1,1.5" target="0,0,5" up="0,1,0"/></transform></emitter>
  <emitter type="projector" id="Projector"><float name="scale" value="20"/>
    <transform name="to_world"><lookat origin="0.25,0,1.5" target="0,0,5" up="0,1,0"/></transform></emitter>
</scene>"""


def test_obj_roundtrip_and_polygons(tmp_path):
    v, t = scenes.make_uv_sphere((0, 0, 0), 1.0, 8, 4)
    loaders.save_obj(tmp_path / "s.obj", v, t)
    v2, t2 = loaders.load_obj(tmp_path / "s.obj")
    np.testing.assert_allclose(v2, v, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(t2, t)
    (tmp_path / "p.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\nf -4 -3 -2\n")
    v3, t3 = loaders.load_obj(tmp_path / "p.obj")
    assert v3.shape == (4, 3)
    np.testing.assert_array_equal(t3, [[0, 1, 2], [0, 2, 3], [0, 1, 2]])
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        loaders.load_obj(tmp_path / "bad.obj")


def test_obj_sequence(tmp_path):
    v, t = scenes.make_plane(1.0, 1.0, 2, 2)
    for k in range(3):
        loaders.save_obj(tmp_path / f"f{k:03d}.obj", v + k, t)
    frames, tris = loaders.load_obj_sequence(str(tmp_path))
    assert frames.shape == (3, 9, 3) and tris.shape == (8, 3)
    np.testing.assert_allclose(frames[2], v + 2, rtol=1e-6)
    # the Mesh API reads the same files (mesh.py:167-181)
    import torch

    import fireflies_amd as ff

    m = ff.entity.Mesh("m", torch.from_numpy(v), "cpu")
    m.add_train_animation_from_obj(str(tmp_path))
    m.add_eval_animation_from_obj(str(tmp_path))
    assert tuple(m._anim_data_train.shape) == (3, 9, 3) and m.animated()


def test_mitsuba_xml_subset(tmp_path):
    wv, wt = scenes.make_plane(0.0, 1.0, 2, 2)
    qv, qt = scenes.make_plane(4.0, 0.3, 1, 1)
    loaders.save_obj(tmp_path / "wall.obj", wv, wt)
    loaders.save_obj(tmp_path / "quad.obj", qv, qt)
    (tmp_path / "scene.xml").write_text(XML)
    sc = loaders.load_mitsuba_xml(str(tmp_path / "scene.xml"))
    assert [m.name for m in sc.meshes] == ["mesh-Wall", "mesh-Quad"]
    # to_world = translate(z=6) . scale(2), in document order
    np.testing.assert_allclose(sc.meshes[0].frames[0], wv * 2 + np.array([0, 0, 6], np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sc.meshes[0].albedo, (0.8, 0.3, 0.35))
    assert sc.meshes[0].material == "mat-Mucosa" and sc.meshes[1].albedo == (0.2, 0.2, 0.2)
    # the principled material keeps its scalar parameters; the diffuse one has none
    assert sc.meshes[0].bsdf == {"roughness": 0.35, "specular": 0.6, "clearcoat": 0.25} and sc.meshes[1].bsdf is None
    rows = scenes.material_rows(sc)
    assert rows.shape == (2, 16) and rows[0, 3] == 1.0 and rows[1, 3] == 0.0
    np.testing.assert_allclose(rows[0, [4, 13]], [0.35, 0.25], rtol=1e-6)
    np.testing.assert_allclose(rows[0, 8], 2.0 / (1.0 - np.sqrt(0.08 * 0.6)) - 1.0, rtol=1e-6)
    np.testing.assert_allclose(rows[1, :3], 0.2, rtol=1e-6)
    assert (sc.camera.width, sc.camera.height, sc.camera.fov_x) == (96, 96, 60.0)
    np.testing.assert_allclose(sc.camera.to_world, scenes.look_at((0, 0, 1.5), (0, 0, 5)), atol=1e-6)
    assert sc.projector.name == "PerspectiveCamera_1" and (sc.projector.width, sc.projector.fov_x) == (128, 30.0)
    np.testing.assert_allclose(sc.projector.to_world, scenes.look_at((0.25, 0, 1.5), (0, 0, 5)), atol=1e-6)
    assert sc.projector_scale == 20.0
    assert sc.spot.intensity == (8.0, 8.0, 8.0) and sc.spot.cutoff_angle == 40.0 and sc.spot.beam_width == 30.0
    # rotate: 90 degrees about y maps +x to -z
    import xml.etree.ElementTree as ET

    M = loaders._transform(ET.fromstring('<transform><rotate y="1" angle="90"/></transform>'))
    np.testing.assert_allclose(M[:3, :3] @ np.array([1.0, 0, 0]), [0, 0, -1], atol=1e-6)


def test_ply_ascii_binary_and_polygons(tmp_path):
    """PLY meshes (Scene classifies "ply" keys as meshes like the reference, fireflies/scene.py:100; Mitsuba scene
    files reference them as <shape type="ply">): ascii and both binary byte orders, extra vertex properties,
    polygon faces, a leading element that is neither vertex nor face."""
    v, t = scenes.make_uv_sphere((0.1, -0.2, 0.3), 0.7, 10, 6)
    for binary in (True, False):
        f = tmp_path / f"s{int(binary)}.ply"
        loaders.save_ply(f, v, t, binary=binary)
        v2, t2 = loaders.load_ply(f)
        np.testing.assert_allclose(v2, v, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(t2, t)
        v3, t3 = loaders.load_mesh_file(str(f))
        assert v3.shape == v.shape and t3.shape == t.shape
    # hand-written ascii file: normals + colours on the vertices, a quad and a pentagon, a comment, another element first
    (tmp_path / "q.ply").write_text(
        "ply\nformat ascii 1.0\ncomment made by hand\nelement material 1\nproperty float shininess\nelement vertex 5\nproperty float x\nproperty float y\n"
        "property float z\nproperty float nx\nproperty uchar red\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n"
        "0.5\n0 0 0 0 255\n1 0 0 0 255\n1 1 0 0 255\n0 1 0 0 255\n0.5 1.5 0 0 255\n4 0 1 2 3\n5 0 1 2 4 3\n")
    vq, tq = loaders.load_ply(tmp_path / "q.ply")
    assert vq.shape == (5, 3) and tq.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 4], [0, 4, 3]]
    # big-endian binary with double coordinates and ushort indices
    import struct

    head = b"ply\nformat binary_big_endian 1.0\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\nelement face 1\nproperty list uchar ushort vertex_index\nend_header\n"
    body = b"".join(struct.pack(">ddd", *p) for p in ((0, 0, 0), (1, 0, 0), (0, 2, 0))) + struct.pack(">BHHH", 3, 0, 1, 2)
    (tmp_path / "b.ply").write_bytes(head + body)
    vb, tb = loaders.load_ply(tmp_path / "b.ply")
    np.testing.assert_allclose(vb, [[0, 0, 0], [1, 0, 0], [0, 2, 0]])
    assert tb.tolist() == [[0, 1, 2]]
    (tmp_path / "bad.ply").write_text("plx\n")
    with pytest.raises(ValueError):
        loaders.load_ply(tmp_path / "bad.ply")
    # a PLY shape inside a scene file
    wv, wt = scenes.make_plane(0.0, 1.0, 4, 4)
    loaders.save_ply(tmp_path / "wall.ply", wv, wt)
    loaders.save_obj(tmp_path / "quad.obj", *scenes.make_uv_sphere((0.1, 0.05, 4.0), 0.35, 8, 4))
    (tmp_path / "scene.xml").write_text(XML.replace('<shape type="obj" id="mesh-Wall">', '<shape type="ply" id="mesh-Wall">').replace("wall.obj", "wall.ply"))
    sc = loaders.load_mitsuba_xml(str(tmp_path / "scene.xml"))
    assert sc.meshes[0].name == "mesh-Wall" and sc.meshes[0].tris.shape[0] == wt.shape[0]


def test_nothing_the_file_asks_for_is_dropped_silently(tmp_path):
    """Every node or property of a scene file that the reader does not honour is reported (round 2 dropped <integrator>,
    <rfilter>, <sampler> and fov_axis without a word): deep integrators, Mitsuba's default gaussian film filter, other
    samplers, unknown top-level nodes and properties, area emitters, textures, spec_trans.  A file that asks only for what
    is implemented (path with max_depth 2, box or gaussian filter, independent sampler) loads without any warning."""
    import warnings

    wv, wt = scenes.make_plane(0.0, 1.0, 2, 2)
    qv, qt = scenes.make_plane(4.0, 0.3, 1, 1)
    loaders.save_obj(tmp_path / "wall.obj", wv, wt)
    loaders.save_obj(tmp_path / "quad.obj", qv, qt)

    def load(xml):
        (tmp_path / "s.xml").write_text(xml)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sc = loaders.load_mitsuba_xml(str(tmp_path / "s.xml"))
        return sc, [str(x.message) for x in w]

    clean = (XML.replace('<sensor type="perspective" id="PerspectiveCamera">', '<integrator type="path"><integer name="max_depth" value="2"/></integrator>\n'
                         '<sensor type="perspective" id="PerspectiveCamera"><sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>')
             .replace('<integer name="height" value="$res"/></film>', '<integer name="height" value="$res"/><rfilter type="box"/></film>'))
    sc, msgs = load(clean)
    assert msgs == [], msgs
    assert sc.notes["integrator"] == {"type": "path", "max_depth": 2} and sc.notes["rfilter"] == "box" and sc.notes["sample_count"] == 64
    # the plain file: Mitsuba's defaults are a gaussian film filter and an unbounded path integrator
    # (the gaussian filter is implemented since round 4: it is selected — notes["rfilter"], which mi.Scene renders with — not reported)
    sc_plain, msgs = load(XML)
    assert any("no <integrator>" in m for m in msgs) and not any("reconstruction filter" in m for m in msgs)
    assert sc_plain.notes["rfilter"] == "gaussian" and sc_plain.notes["rfilter_stddev"] == 0.5
    sc_g, msgs = load(clean.replace('<rfilter type="box"/>', '<rfilter type="gaussian"><float name="stddev" value="0.4"/></rfilter>'))
    assert msgs == [] and sc_g.notes["rfilter"] == "gaussian" and sc_g.notes["rfilter_stddev"] == pytest.approx(0.4)
    _, msgs = load(clean.replace('<rfilter type="box"/>', '<rfilter type="gaussian"><float name="stddev" value="1.0"/></rfilter>'))
    assert any("stddev 1.0" in m and "5x5" in m for m in msgs)
    # what round 2 swallowed
    noisy = (clean.replace('value="2"/></integrator>', 'value="8"/></integrator>\n<medium type="homogeneous"/>')
             .replace('<rfilter type="box"/>', '<rfilter type="tent"/>')
             .replace('<sampler type="independent">', '<sampler type="stratified">')
             .replace('<float name="fov" value="60"/>', '<float name="fov" value="60"/><string name="fov_axis" value="y"/><float name="aperture_radius" value="0.1"/>')
             .replace('<float name="clearcoat" value="0.25"/>', '<float name="clearcoat" value="0.25"/><float name="spec_trans" value="0.3"/>'
                      '<texture type="bitmap" name="base_color"><string name="filename" value="t.png"/></texture>'
                      '<texture type="bitmap" name="roughness"><string name="filename" value="r.png"/></texture>')
             .replace('<bsdf type="diffuse"><rgb name="reflectance" value="0.2"/></bsdf></shape>',
                      '<bsdf type="diffuse"><rgb name="reflectance" value="0.2"/></bsdf><emitter type="area"><rgb name="radiance" value="1"/></emitter></shape>\n'
                      '<emitter type="envmap"><string name="filename" value="e.exr"/></emitter>'))
    sc, msgs = load(noisy)
    for needle in ("max_depth 8", "top-level <medium> ignored", "reconstruction filter 'tent'", "sampler type 'stratified'", "sensor property 'aperture_radius' is ignored",
                   "spec_trans > 0", "texture-valued parameter 'roughness' is not evaluated", "base-colour bitmap 't.png' could not be read",
                   "a textured base colour but no texture coordinates", "area emitter on a shape ignored", "emitter type 'envmap' ignored"):
        assert any(needle in m for m in msgs), (needle, msgs)
    # fov_axis = y on a square film is the same angle; on a 4:3 film it is converted to the horizontal angle
    assert sc.camera.fov_x == pytest.approx(60.0)
    assert loaders._fov_x(45.0, "y", 400, 300) == pytest.approx(np.rad2deg(2 * np.arctan(np.tan(np.deg2rad(22.5)) * 4 / 3)))
    assert loaders._fov_x(45.0, "smaller", 400, 300) == loaders._fov_x(45.0, "y", 400, 300) and loaders._fov_x(45.0, "larger", 400, 300) == 45.0
    assert loaders._fov_x(50.0, "diagonal", 300, 400) == pytest.approx(np.rad2deg(2 * np.arctan(np.tan(np.deg2rad(25.0)) * 0.6)))
    other, msgs = load(clean.replace('<integrator type="path">', '<integrator type="ptracer">'))
    assert any("'ptracer'" in m and "not implemented" in m for m in msgs)


def test_obj_texture_coordinates_and_a_bitmap_base_colour(tmp_path):
    """`vt` records become per-vertex texture coordinates (v flipped like Mitsuba's flip_tex_coords default); a vertex used with
    two different `vt` (a seam) is duplicated; a <texture type="bitmap" name="base_color"> inside the bsdf gives the mesh a
    texture-valued base colour (read with PIL when present) — the scene then exposes `<mat>.brdf_0.base_color.data`."""
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                                    "f 1/1 2/2 3/3\nf 1/5 3/3 4/4\n")  # vertex 1 carries vt 1 in one face and vt 5 in the other
    v, t, info = loaders.load_obj(tmp_path / "q.obj", with_info=True)
    assert v.shape == (5, 3) and t.tolist() == [[0, 1, 2], [4, 2, 3]] and not info["has_normals"]
    np.testing.assert_allclose(v[4], v[0])
    np.testing.assert_allclose(info["uv"], [[0, 1], [1, 1], [1, 0], [0, 0], [0.5, 0.5]])
    v2, t2 = loaders.load_obj(tmp_path / "q.obj")  # the plain call keeps the file's vertices
    assert v2.shape == (4, 3)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        px = np.zeros((2, 3, 3), np.uint8)
        px[0, 1] = (255, 128, 0)
        Image.fromarray(px).save(tmp_path / "t.png")
    wv, wt = scenes.make_plane(0.0, 1.0, 2, 2)
    loaders.save_obj(tmp_path / "wall.obj", wv, wt)
    xml = XML.replace("wall.obj", "q.obj").replace('<float name="clearcoat" value="0.25"/>',
                                                   '<float name="clearcoat" value="0.25"/><texture type="bitmap" name="base_color"><string name="filename" value="t.png"/></texture>')
    loaders.save_obj(tmp_path / "quad.obj", *scenes.make_plane(4.0, 0.3, 1, 1))
    (tmp_path / "s.xml").write_text(xml)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = loaders.load_mitsuba_xml(str(tmp_path / "s.xml"))
    m = sc.meshes[0]
    assert m.uv.shape == (5, 2) and m.base_tex is not None and m.base_tex.shape[-1] == 3 and sc.meshes[1].base_tex is None
    if Image is not None:
        assert m.base_tex.shape == (2, 3, 3)
        np.testing.assert_allclose(m.base_tex[0, 1], [1.0, ((128 / 255 + 0.055) / 1.055) ** 2.4, 0.0], rtol=1e-5)  # sRGB -> linear
    rows = scenes.material_rows(sc)
    assert rows[0, 15] == 1.0 and rows[1, 15] == 0.0 and [n for n, _ in scenes.base_textures(sc)] == ["mat-Mucosa"]
    suv = scenes.slot_uv_table(np.array([1, 0, 2, 3]), np.concatenate([m.tris, sc.meshes[1].tris]), np.array([0, 0, 1, 1]), sc.meshes)
    np.testing.assert_allclose(suv[0], m.uv[m.tris[1]].reshape(-1))
    assert suv.shape == (8, 6) and (suv[2:] == 0).all()


def test_seam_vertices_are_only_split_for_a_textured_base_colour(tmp_path):
    """A Blender-style OBJ with UV seams but NO texture bound to its material keeps the file's vertex list: OBJ animation frames and user
    `vertex_positions` (fireflies/entity/mesh.py:167-181 feeds the raw OBJ vertex list) carry V vertices and must still fit the mesh
    (round-3 advisor finding: the loader split seams for every XML shape and `mi.Scene._set_pose` then refused the frames)."""
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                                    "f 1/1 2/2 3/3\nf 1/5 3/3 4/4\n")
    v, t, info = loaders.load_obj(tmp_path / "q.obj", with_info=True, split_seams=False)
    assert v.shape == (4, 3) and t.tolist() == [[0, 1, 2], [0, 2, 3]] and info["uv"] is None
    loaders.save_obj(tmp_path / "quad.obj", *scenes.make_plane(4.0, 0.3, 1, 1))
    (tmp_path / "s.xml").write_text(XML.replace("wall.obj", "q.obj"))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = loaders.load_mitsuba_xml(str(tmp_path / "s.xml"))
    m = sc.meshes[0]
    assert m.frames.shape[1] == 4 and m.uv is None and m.base_tex is None  # the file's four vertices: frames of the same OBJ topology fit
