"""The C++ host side (misaki-render_amd/host): the reference's XML / Properties / plugin interface.
CPU tests cover loading, flattening and error behaviour; the GPU test renders through
scene->integrator()->render() and compares with the oracle."""
import importlib
import os
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def hostlib():
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    return importlib.import_module("misaki-render_amd.hostlib")


@pytest.fixture()
def cbox_xml(tmp_path, hostmirror):
    return hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 64, 48, 4)


def test_load_and_flatten_matches_the_python_mirror(hostlib, hostmirror, cbox_xml, oracle):
    sc = hostlib.HostScene(cbox_xml)
    assert sc.film_size() == (64, 48, 4)
    flat = sc.flatten()
    ref = hostmirror.cbox_scene(64, 48)
    d, r = flat.desc, ref.desc
    assert (d.n_meshes, d.n_bsdfs, d.n_emitters, d.n_faces, d.n_vertices) == (8, 8, 1, 32, 64)
    assert np.array_equal(flat.vertices, ref.vertices) and np.array_equal(flat.faces, ref.faces)   # OBJ round trip, quad split
    assert [(d.meshes[i].emitter_id, d.meshes[i].first_face, d.meshes[i].face_count) for i in range(8)] == \
           [(r.meshes[i].emitter_id, r.meshes[i].first_face, r.meshes[i].face_count) for i in range(8)]
    assert d.emitters[0].d65_scale == r.emitters[0].d65_scale and d.emitters[0].mesh_id == 0
    # one flattener truth: camera, filter table and spectral coefficients are the mirror's, bit for bit (more sizes below)
    assert np.array_equal(np.array(d.camera.sample_to_camera[:], np.float32).view(np.uint32), np.array(r.camera.sample_to_camera[:], np.float32).view(np.uint32))
    assert np.array_equal(np.array(d.camera.to_world[:], np.float32).view(np.uint32), np.array(r.camera.to_world[:], np.float32).view(np.uint32))
    assert d.camera.near_clip == 10.0 and d.camera.far_clip == 2800.0
    assert d.film.filter_radius == 2.0 and np.array_equal(np.array(d.film.filter_lut[:], np.float32).view(np.uint32), np.array(r.film.filter_lut[:], np.float32).view(np.uint32))
    assert np.array_equal(np.ctypeslib.as_array(d.cie1931_xyz, (285,)), np.ctypeslib.as_array(r.cie1931_xyz, (285,)))
    for i in range(8):
        assert np.array_equal(np.array(d.bsdfs[i].reflectance[:], np.float32).view(np.uint32), np.array(r.bsdfs[i].reflectance[:], np.float32).view(np.uint32))
    assert np.array_equal(np.array(d.emitters[0].radiance[:], np.float32).view(np.uint32), np.array(r.emitters[0].radiance[:], np.float32).view(np.uint32))
    # render parameters = the reference's effective integrator settings
    p = flat.params
    assert (p.spp, p.rng_mode, p.rr_depth, p.max_depth, p.hide_emitters, p.block_size) == (4, 1, 5, -1, 0, 32)
    # the flattened scene is a valid input of the oracle (and therefore of the C ABI)
    osc = oracle.scene(flat)
    film, st = osc.render(p, threads=2)
    assert st.samples == 64 * 48 * 4 and np.isfinite(film).all() and film[..., 4].min() > 0
    osc.close()
    sc.close()


@pytest.mark.parametrize("w,h", [(64, 48), (800, 600), (1920, 1080), (100, 40), (512, 512)])
def test_the_two_flatteners_produce_the_same_bits(hostlib, hostmirror, tmp_path, w, h):
    """ONE flattener truth at the boundary: the C++ host (XML -> plugins -> flatten: what the drop-in hands to the C ABI) and the
    python mirror (what the parity suite and bench.py build their scenes with) give bit-identical camera matrices — also at
    aspect ratios != 1, where round 4's two arithmetics parted in the last place —, filter tables, spectral coefficients,
    geometry and tables, so a film rendered through either is the same film."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), w, h, 4)
    sc = hostlib.HostScene(xml)
    flat, ref = sc.flatten(), hostmirror.cbox_scene(w, h)
    d, r = flat.desc, ref.desc
    bits = lambda a: np.array(a[:], np.float32).view(np.uint32)
    assert np.array_equal(bits(d.camera.sample_to_camera), bits(r.camera.sample_to_camera))
    assert np.array_equal(bits(d.camera.to_world), bits(r.camera.to_world))
    assert (d.camera.near_clip, d.camera.far_clip) == (r.camera.near_clip, r.camera.far_clip)
    assert d.film.filter_radius == r.film.filter_radius and np.array_equal(bits(d.film.filter_lut), bits(r.film.filter_lut))
    assert (d.film.width, d.film.height) == (r.film.width, r.film.height) == (w, h)
    assert np.array_equal(flat.vertices.view(np.uint32), ref.vertices.view(np.uint32)) and np.array_equal(flat.faces, ref.faces)
    assert d.n_bsdfs == r.n_bsdfs and d.n_emitters == r.n_emitters
    for i in range(d.n_bsdfs):
        assert np.array_equal(bits(d.bsdfs[i].reflectance), bits(r.bsdfs[i].reflectance)), i
        assert bytes(d.bsdfs[i]) == bytes(r.bsdfs[i]), i
    for i in range(d.n_emitters):
        assert bytes(d.emitters[i]) == bytes(r.emitters[i]), i
    assert bytes(d.camera) == bytes(r.camera)
    # (the film's crop window: the plugin always writes it out, the mirror leaves {0, 0} = the whole film: the same window)
    assert tuple(d.film.crop_offset) == (0, 0) and tuple(d.film.crop_size) == (w, h) and tuple(r.film.crop_size) in ((0, 0), (w, h))
    sc.close()


def test_parameters_and_properties(hostlib, hostmirror, tmp_path):
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 32, 32, 2,
                                     integrator_props={"max_depth": 3, "rr_depth": 2, "honor_properties": True, "block_size": 16},
                                     film_type="rgbfilm")
    sc = hostlib.HostScene(xml, spp=7, width=40)          # $spp / $width override the <default>s
    assert sc.film_size() == (40, 32, 7)
    p = sc.flatten().params
    assert (p.max_depth, p.rr_depth, p.block_size) == (3, 2, 16)
    sc.close()


def test_error_behaviour(hostlib, hostmirror, tmp_path, cbox_xml):
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(str(tmp_path / "nope.xml"))
    assert "file not exists" in str(e.value)
    bad = open(cbox_xml).read()
    cases = {
        "unknown plugin": (bad.replace('bsdf type="diffuse"', 'bsdf type="velvet"', 1), 'Plugin "velvet" not found'),
        "bad rr_depth": (bad.replace('<integrator type="path">', '<integrator type="path"><integer name="rr_depth" value="0"/>'),
                         '"rr_depth" must be set to a value greater than zero!'),
        "unexpected tag": (bad.replace("<scene>", "<scene><bogus/>", 1), 'unexpected tag "bogus"'),
        "missing mesh": (bad.replace("meshes/cbox_floor.obj", "meshes/missing.obj"), "file not found"),
        "wrong type": (bad.replace('<float name="fov"', '<string name="fov"'), 'wrong type'),
        "unterminated": (bad.replace("</scene>", ""), "missing closing tag"),
        "two cameras": (bad.replace("</scene>", '<sensor type="perspective"/></scene>'), "Can only have one camera."),
    }
    for name, (text, needle) in cases.items():
        p = tmp_path / (name.replace(" ", "_") + ".xml")
        p.write_text(text)
        with pytest.raises(hostlib.HostError) as e:
            hostlib.HostScene(str(p)).flatten()
        assert needle in str(e.value), (name, str(e.value))


def test_transform_ops_and_default_bsdf(hostlib, hostmirror, tmp_path):
    """<rotate>/<scale>/<translate> compose right to left; a shape without <bsdf> gets diffuse(0.5)."""
    m = hostmirror.MeshSpec("tri", [((0, 0, 0), (1, 0, 0), (0, 1, 0))], (0.5, 0.5, 0.5))
    xml = hostmirror.write_scene_xml([m], str(tmp_path), 16, 16, 1)
    text = open(xml).read().replace('<bsdf type="diffuse">\n            <rgb name="reflectance" value="0.5, 0.5, 0.5"/>\n        </bsdf>',
                                    '<transform name="to_world"><scale value="2"/><rotate z="1" angle="90"/><translate x="1" y="2" z="3"/></transform>')
    assert "<rotate" in text
    (tmp_path / "t.xml").write_text(text)
    flat = hostlib.HostScene(str(tmp_path / "t.xml")).flatten()
    v = flat.vertices[:, :3]
    assert np.allclose(v, [[1, 2, 3], [1, 4, 3], [-1, 2, 3]], atol=1e-5)      # scale 2 -> rotate 90 about z -> translate
    # grey 0.5 -> the reference table's (not exactly flat) entry
    assert flat.desc.n_bsdfs == 1 and np.allclose(flat.desc.bsdfs[0].reflectance[:], [-2.2974573e-09, 1.5341052e-06, -1.3818033e-04], rtol=1e-6, atol=0)


def test_roughconductor_and_twosided_plugins(hostlib, hostmirror, tmp_path, abi):
    """bsdfs/roughconductor.cpp:12-50 and bsdfs/twosided.cpp:12-36 through the XML loader."""
    m = hostmirror.MeshSpec("tri", [((0, 0, 0), (1, 0, 0), (0, 1, 0))], (0.5, 0.5, 0.5))
    xml = hostmirror.write_scene_xml([m, m], str(tmp_path), 16, 16, 1)
    text = open(xml).read()
    plain = '<bsdf type="diffuse">\n            <rgb name="reflectance" value="0.5, 0.5, 0.5"/>\n        </bsdf>'
    rc = ('<bsdf type="roughconductor"><float name="alpha" value="0.25"/><string name="distribution" value="ggx"/>'
          '<rgb name="eta" value="2.8656, 2.11918, 1.94008"/><rgb name="k" value="3.03233, 2.05611, 1.61629"/></bsdf>')
    two = '<bsdf type="twosided">' + rc.replace('value="0.25"', 'value="0.1"') + '<bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.3, 0.4"/></bsdf></bsdf>'
    text = text.replace(plain, rc, 1).replace(plain, two, 1)
    (tmp_path / "rc.xml").write_text(text)
    d = hostlib.HostScene(str(tmp_path / "rc.xml")).flatten().desc
    assert d.n_bsdfs == 3                       # rough conductor | twosided back (diffuse) | twosided front
    b0 = d.bsdfs[d.meshes[0].bsdf_id]
    assert b0.type == abi.MSK_BSDF_ROUGHCONDUCTOR and b0.back_bsdf == -1 and b0.alpha_u == b0.alpha_v == np.float32(0.25)
    assert np.isclose(b0.eta.scale, 2 * 2.8656) and np.isclose(b0.k.scale, 2 * 3.03233) and b0.specular_reflectance.scale == 1.0
    # the default specular_reflectance is srgb(1,1,1) (properties.cpp:226-235 with the key fixed): the table's white
    assert np.allclose(b0.specular_reflectance.coeff[:], [0.0009053870453499258, -1.055624008178711, 309.9350280761719], rtol=1e-6)
    b1 = d.bsdfs[d.meshes[1].bsdf_id]
    back = d.bsdfs[b1.back_bsdf]
    assert b1.type == abi.MSK_BSDF_ROUGHCONDUCTOR and b1.alpha_u == np.float32(0.1) and back.type == abi.MSK_BSDF_DIFFUSE
    for bad, needle in ((rc.replace("ggx", "beckmann"), "beckmann"), (rc.replace('<rgb name="k" value="3.03233, 2.05611, 1.61629"/>', ""), "eta"),
                        ('<bsdf type="twosided"></bsdf>', "A nested one-sided material is required!")):
        (tmp_path / "bad.xml").write_text(open(xml).read().replace(plain, bad, 1))
        with pytest.raises(hostlib.HostError) as e:
            hostlib.HostScene(str(tmp_path / "bad.xml"))
        assert needle in str(e.value)


def test_image_writers(hostlib, tmp_path):
    img = np.random.RandomState(0).rand(5, 7, 4).astype(np.float32)
    hostlib.write_image(tmp_path / "a.exr", img)
    raw = (tmp_path / "a.exr").read_bytes()
    assert struct.unpack("<I", raw[:4])[0] == 20000630 and b"channels\0chlist\0" in raw and b"compression\0" in raw
    # uncompressed scanlines: the last row's R plane (channels stored alphabetically A,B,G,R) ends the file
    assert np.array_equal(np.frombuffer(raw[-7 * 4:], np.float32), img[-1, :, 0])
    hostlib.write_image(tmp_path / "a.pfm", img[..., :3].copy())
    head, rest = (tmp_path / "a.pfm").read_bytes().split(b"-1.0\n", 1)
    assert head == b"PF\n7 5\n" and np.array_equal(np.frombuffer(rest, np.float32).reshape(5, 7, 3)[::-1], img[..., :3])


@pytest.mark.gpu
def test_render_through_the_plugin_interface(hostlib, hostmirror, oracle, tmp_path):
    """scene->integrator()->render(scene, sensor) -> Film::put -> HDRFilm::image(): bit-identical to the oracle."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 96, 64, 8)
    sc = hostlib.HostScene(xml)
    film, rgba, st = sc.render(develop_to=str(tmp_path / "out.exr"))
    flat = sc.flatten()
    ref, rst = oracle.scene(flat).render(flat.params, threads=4)
    assert st.samples == rst.samples == 96 * 64 * 8
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    assert np.allclose(rgba, hostmirror.develop(ref), rtol=1e-6, atol=1e-7)
    assert os.path.getsize(tmp_path / "out.exr") > 96 * 64 * 16
    sc.close()


def test_film_crop_window_properties(hostlib, hostmirror, tmp_path):
    """<film> crop_offset_x / _y, crop_width / _height (film.cpp:12-21): flattened into msk_film_desc; the reference's check and
    message (film.cpp:51-63) for a window outside the film."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 2,
                                     film_props={"crop_offset_x": 11, "crop_offset_y": 5, "crop_width": 37, "crop_height": 21})
    sc = hostlib.HostScene(xml)
    assert sc.film_size()[:2] == (100, 40) and sc.film_crop() == (11, 5, 37, 21)
    f = sc.flatten().desc.film
    assert (f.width, f.height, tuple(f.crop_offset), tuple(f.crop_size)) == (100, 40, (11, 5), (37, 21))
    sc.close()
    plain = hostlib.HostScene(hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 2, filename="plain.xml"))
    assert plain.film_crop() == (0, 0, 100, 40) and tuple(plain.flatten().desc.film.crop_size) == (100, 40)
    plain.close()
    bad = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 2, filename="bad.xml",
                                     film_props={"crop_offset_x": 80, "crop_width": 37})
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(bad)
    assert "Invalid crop window specification" in str(e.value)


@pytest.mark.gpu
def test_render_a_crop_window_through_the_plugin(hostlib, hostmirror, oracle, tmp_path):
    """The "path" plugin with a cropped film: one borderless ImageBlock of the crop size at the crop offset goes to Film::put
    (hdrfilm.cpp:37-46), HDRFilm::image() develops that window — the oracle's film, bit for bit."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 4,
                                     film_props={"crop_offset_x": 11, "crop_offset_y": 5, "crop_width": 37, "crop_height": 21})
    sc = hostlib.HostScene(xml)
    film, rgba, st = sc.render(develop_to=str(tmp_path / "crop.exr"))
    flat = sc.flatten()
    ref, rst = oracle.scene(flat).render(flat.params, threads=4)
    assert film.shape == (21, 37, 5) and rgba.shape == (21, 37, 4) and st.samples == rst.samples
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    assert np.allclose(rgba, hostmirror.develop(ref), rtol=1e-6, atol=1e-7)
    sc.close()


def test_rng_property_of_the_path_plugin(hostlib, hostmirror, tmp_path, abi):
    """<integrator type="path"><string name="rng" value="pcg_block"/>: the reference's sampler semantics (one PCG32 stream per
    block) instead of the counter RNG; anything else is refused when the scene is loaded."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 64, 48, 2, integrator_props={"rng": "pcg_block"})
    sc = hostlib.HostScene(xml)
    assert sc.flatten().params.rng_mode == abi.MSK_RNG_PCG_BLOCK
    sc.close()
    (tmp_path / "bad_rng.xml").write_text(open(xml).read().replace('value="pcg_block"', 'value="sobol"'))
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(str(tmp_path / "bad_rng.xml"))
    assert '"rng" must be' in str(e.value)


@pytest.mark.gpu
def test_plugin_renders_in_the_references_sampler_mode(hostlib, hostmirror, oracle, tmp_path):
    """The "path" plugin with rng="pcg_block": scene->integrator()->render() on the GPU == the oracle's pcg_block film, bit for bit."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 96, 64, 4, integrator_props={"rng": "pcg_block"})
    sc = hostlib.HostScene(xml)
    film, rgba, st = sc.render()
    flat = sc.flatten()
    ref, rst = oracle.scene(flat).render(flat.params, threads=4)
    assert st.samples == rst.samples == 96 * 64 * 4 and st.segments == rst.segments
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    sc.close()


@pytest.mark.gpu
def test_plugin_gpu_devices_property(hostlib, hostmirror, oracle, tmp_path):
    """<integrator type="path"><string name="gpu_devices" value="0,0"/>: the plugin hands both ordinals to msk_gpu_init, the
    library shards the samples over two member contexts and sums the films (msk_multi.h) — rehearsed on one GPU."""
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 64, 48, 6, integrator_props={"gpu_devices": "0,0"})
    assert 'name="gpu_devices"' in open(xml).read()
    sc = hostlib.HostScene(xml)
    film, rgba, st = sc.render()
    flat = sc.flatten()
    ref, rst = oracle.scene(flat).render(flat.params, threads=4)
    assert st.samples == rst.samples == 64 * 48 * 6 and st.segments == rst.segments
    assert np.allclose(film, ref, rtol=2e-6, atol=1e-6)
    assert np.abs(rgba[..., :3] - hostmirror.develop(ref)[..., :3]).max() < 1e-4
    sc.close()
    bad = open(xml).read().replace('value="0,0"', 'value="0,x"')
    (tmp_path / "bad.xml").write_text(bad)
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(str(tmp_path / "bad.xml"))
    assert "gpu_devices" in str(e.value)


def _cli():
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    return os.path.join(ge.PKG, "lib", "misaki-cli")


def test_cli_usage_and_errors(cbox_xml, tmp_path):
    """misaki-cli (the reference's src/apps/main.cpp with arguments): usage, help, and the caught-exception path
    (main.cpp:55-57) — none of which touches the GPU."""
    import subprocess
    cli = _cli()
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: misaki-cli" in r.stderr
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout
    r = subprocess.run([cli, str(tmp_path / "missing.xml")], capture_output=True, text=True)
    assert r.returncode == 1 and "Caught a critical exception" in (r.stderr + r.stdout) and "file not exists" in (r.stderr + r.stdout)
    bad = tmp_path / "bad.xml"
    bad.write_text(open(cbox_xml).read().replace('<integrator type="path">', '<integrator type="path"><integer name="rr_depth" value="0"/>'))
    r = subprocess.run([cli, str(bad), "-q"], capture_output=True, text=True)
    assert r.returncode == 1 and "rr_depth" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_cli_renders_a_scene_file_to_exr(hostmirror, tmp_path):
    import subprocess
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 64, 48, 4)
    out = tmp_path / "image.exr"
    r = subprocess.run([_cli(), xml, "-o", str(out), "-D", "spp=2", "-q"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = out.read_bytes()
    assert struct.unpack("<I", raw[:4])[0] == 20000630 and len(raw) > 64 * 48 * 16
    # the last scanline's R plane is finite and the image is not black
    tail = np.frombuffer(raw[-64 * 4:], np.float32)
    assert np.isfinite(tail).all() and np.frombuffer(raw[-64 * 48 * 16:], np.float32).max() > 0
