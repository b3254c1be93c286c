"""Scene-file readers: OBJ and the Mitsuba-XML subset.  Host side only."""
import numpy as np
import pytest

from fireflies_amd import loaders, scenes

XML = """<scene version="3.0.0">
  <default name="spp" value="64"/>
  <default name="res" value="96"/>
  <bsdf type="twosided" id="mat-Mucosa"><bsdf type="principled"><rgb name="base_color" value="0.8, 0.3, 0.35"/></bsdf></bsdf>
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
