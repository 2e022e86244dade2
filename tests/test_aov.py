"""the "aov" integrator (integrators/aov.cpp:87-144 through SamplingIntegrator::render_sample,
integrator.cpp:103-126): primary-hit channels and the nested path integrator's RGBA ride the same film filter."""
import importlib

import numpy as np
import pytest


def aov_scene(hostmirror, w, h):
    meshes = hostmirror.cbox_meshes()
    # a blob with vertex normals would need an OBJ with vn; the flat description carries them directly
    return hostmirror.flatten(meshes, w, h)


def test_channel_layout_and_primary_hit_values(oracle, hostmirror, abi):
    A = abi
    flat = aov_scene(hostmirror, 40, 40)
    sc = oracle.scene(flat)
    types = [A.MSK_AOV_DEPTH, A.MSK_AOV_POSITION, A.MSK_AOV_PATH_RGBA, A.MSK_AOV_GEO_NORMAL, A.MSK_AOV_UV, A.MSK_AOV_SH_NORMAL]
    prm = abi.render_params(4, seed=3)
    film, st = sc.render_aov(prm, types)
    assert film.shape == (40, 40, 5 + 1 + 3 + 4 + 3 + 2 + 3)
    plain, _ = sc.render(prm)
    assert np.array_equal(film[..., :5], plain)                       # the nested path sample is the XYZ result (aov.cpp:138-139)
    w = film[..., 4:5]
    val = film[..., 5:] / w
    depth, pos, rgba, ng, uv, ns = val[..., 0], val[..., 1:4], val[..., 4:8], val[..., 8:11], val[..., 11:13], val[..., 13:16]
    # pixels that only see the back wall: z = 559.2, normal (0,0,-1), depth = distance from the camera at z = -800
    c = np.abs(pos[..., 2] - 559.2) < 1e-3
    assert c.sum() > 50 and np.allclose(ng[c], [0, 0, -1], atol=1e-5) and np.array_equal(ng, ns)
    assert np.all(depth[c] > 1359.2 - 1e-2) and np.all(depth[c] < 1500)
    assert np.allclose(rgba[..., 3], 1.0, rtol=1e-6)                                         # A / W
    # the RGB channels are the developed image WITHOUT the sensor's ray weight (aov.cpp:128-131 converts the nested sample
    # before integrator.cpp:114 multiplies it): the wavelength weights are 1/pdf ~ 254..4380 (spectrum.h:152-181)
    dev = hostmirror.develop(plain)[..., :3]
    assert 200 < dev.mean() / rgba[..., :3].mean() < 500
    assert np.corrcoef(dev.sum(-1).ravel(), rgba[..., :3].sum(-1).ravel())[0, 1] > 0.9
    assert uv.min() >= 0 and uv.max() <= 1 + 1e-6                                             # barycentrics without texcoords
    # positions are consistent with depth along the pixel's camera ray
    cam = np.array([278, 273, -800], np.float32)
    assert np.allclose(np.linalg.norm(pos[c] - cam, axis=-1), depth[c], rtol=2e-3)
    # without a nested integrator XYZ stays 0 and only camera rays are traced
    film2, st2 = sc.render_aov(prm, [A.MSK_AOV_DEPTH])
    assert film2.shape[-1] == 6 and not film2[..., :3].any() and np.array_equal(film2[..., 5], film[..., 5])
    assert st2.shadow_rays == 0 and np.array_equal(film2[..., 3:5], plain[..., 3:5])
    sc.close()


def test_host_plugin_parses_aov_specifications(hostmirror, tmp_path, abi):
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 16, 16, 1)
    text = open(xml).read()
    aov = ('<integrator type="aov"><string name="aovs" value="dd:depth, nn:sh_normal,pp:position,gg:geo_normal,tt:uv"/>'
           '<integrator type="path" name="image"/></integrator>')
    start, end = text.index('<integrator type="path">'), text.index('</integrator>') + len('</integrator>')
    (tmp_path / "aov.xml").write_text(text[:start] + aov + text[end:])
    sc = hostlib.HostScene(str(tmp_path / "aov.xml"))
    assert sc.aov_types() == [abi.MSK_AOV_DEPTH, abi.MSK_AOV_SH_NORMAL, abi.MSK_AOV_POSITION, abi.MSK_AOV_GEO_NORMAL, abi.MSK_AOV_UV,
                              abi.MSK_AOV_PATH_RGBA]
    assert sc.flatten().params.spp == 1
    names = sc.aov_names()
    assert names == ["dd", "nn.X", "nn.Y", "nn.Z", "pp.X", "pp.Y", "pp.Z", "gg.X", "gg.Y", "gg.Z", "tt.U", "tt.V",
                     "image.R", "image.G", "image.B", "image.A"]
    sc.close()
    bad = aov.replace("dd:depth", "dd:albedo")
    (tmp_path / "bad.xml").write_text(text[:start] + bad + text[end:])
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(str(tmp_path / "bad.xml"))
    assert 'Invalid AOV type "albedo"!' in str(e.value)


@pytest.mark.gpu
def test_gpu_aov_film_is_bit_identical_to_the_oracle(gpu_ctx, oracle, hostmirror, abi):
    A = abi
    flat = aov_scene(hostmirror, 96, 80)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    prm = abi.render_params(4, seed=7)
    for types in ([A.MSK_AOV_DEPTH, A.MSK_AOV_POSITION, A.MSK_AOV_PATH_RGBA, A.MSK_AOV_GEO_NORMAL, A.MSK_AOV_UV, A.MSK_AOV_SH_NORMAL],
                  [A.MSK_AOV_PATH_RGBA], [A.MSK_AOV_UV, A.MSK_AOV_DEPTH], []):
        film, st = g.render_aov(prm, types)
        ref, rst = o.render_aov(prm, types)
        assert film.shape == ref.shape
        bad = film.view(np.uint32) != ref.view(np.uint32)
        assert not bad.any(), (types, int(bad.sum()), np.argwhere(bad)[:4])
    # the plain render is unaffected by a preceding AOV render (workspace reuse)
    film, _ = g.render(prm)
    ref, _ = o.render(prm)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(abi.MskError):
        g.render_aov(prm, [A.MSK_AOV_PATH_RGBA, A.MSK_AOV_PATH_RGBA])
    g.close()
    o.close()


@pytest.mark.gpu
def test_gpu_aov_with_normals_texcoords_environment(gpu_ctx, oracle, hostmirror, abi):
    """Smooth normals + texture coordinates (mesh.cpp:68-96) and camera rays that leave the scene."""
    A = abi
    ball = hostmirror.blob_mesh("ball", (278, 273, 280), 150, 16, 24, (0.5, 0.5, 0.5), bump=0.1)
    v, f = hostmirror.triangulate(ball)
    flat = hostmirror.flatten([ball], 64, 64, env={"radiance": None})
    # give the mesh smooth normals (radial) and spherical texcoords in the flat vertex array
    verts = flat.vertices
    d = verts[:, :3] - np.array([278, 273, 280], np.float32)
    n = d / np.linalg.norm(d, axis=1, keepdims=True)
    verts[:, 3:6] = n
    verts[:, 6] = np.arctan2(n[:, 2], n[:, 0]) / (2 * np.pi) + 0.5
    verts[:, 7] = np.arccos(np.clip(n[:, 1], -1, 1)) / np.pi
    flat.desc.meshes[0].has_normals = 1
    flat.desc.meshes[0].has_texcoords = 1
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    prm = abi.render_params(4, seed=1)
    types = [A.MSK_AOV_SH_NORMAL, A.MSK_AOV_UV, A.MSK_AOV_GEO_NORMAL, A.MSK_AOV_PATH_RGBA, A.MSK_AOV_DEPTH]
    film, _ = g.render_aov(prm, types)
    ref, _ = o.render_aov(prm, types)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    val = film[..., 5:] / film[..., 4:5]
    assert not np.array_equal(val[32, 32, 0:3], val[32, 32, 5:8])          # shading normal != geometric normal
    assert not val[2, 2, [0, 1, 2, 3, 4, 5, 6, 7, 12]].any() and val[2, 2, 8:11].min() > 1e-3   # miss: zeros, RGB = environment (without the ray weight)
    g.close()
    o.close()


@pytest.mark.gpu
def test_render_through_the_aov_plugin(oracle, hostmirror, tmp_path, abi):
    """XML -> AOVIntegrator::render -> Film::put -> HDRFilm::image() with the extra channels -> multi-channel EXR."""
    import __graft_entry__ as ge
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 64, 48, 4)
    text = open(xml).read()
    aov = '<integrator type="aov"><string name="aovs" value="dd:depth,nn:sh_normal"/><integrator type="path" name="image"/></integrator>'
    start, end = text.index('<integrator type="path">'), text.index('</integrator>') + len('</integrator>')
    (tmp_path / "aov.xml").write_text(text[:start] + aov + text[end:])
    sc = hostlib.HostScene(str(tmp_path / "aov.xml"))
    film, img, st = sc.render(develop_to=str(tmp_path / "aov.exr"))
    flat = sc.flatten()
    ref, _ = oracle.scene(flat).render_aov(flat.params, sc.aov_types())
    assert film.shape == ref.shape == (48, 64, 5 + 1 + 3 + 4)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    assert np.allclose(img[..., :4], hostmirror.develop(ref[..., :5]), rtol=1e-6, atol=1e-7)
    w = ref[..., 4:5]
    assert np.allclose(img[..., 4:], ref[..., 5:] / w, rtol=1e-6, atol=1e-7)
    raw = (tmp_path / "aov.exr").read_bytes()
    for name in (b"dd\0", b"nn.X\0", b"image.R\0", b"image.A\0"):
        assert name in raw[:2048]
    sc.close()


@pytest.mark.gpu
def test_render_a_crop_window_through_the_aov_plugin(oracle, hostmirror, tmp_path, abi):
    """The "aov" plugin with a cropped film: like "path", ONE block of the crop size at the crop offset goes to Film::put
    (hdrfilm.cpp:37-46) — msk_gpu_render_aov writes crop_h x crop_w x (5 + C) floats, so a full-size block would be clipped
    into a scrambled image (round 4's advisor finding)."""
    import __graft_entry__ as ge
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    xml = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 4,
                                     film_props={"crop_offset_x": 11, "crop_offset_y": 5, "crop_width": 37, "crop_height": 21})
    text = open(xml).read()
    aov = '<integrator type="aov"><string name="aovs" value="dd:depth,pp:position"/><integrator type="path" name="image"/></integrator>'
    start, end = text.index('<integrator type="path">'), text.index('</integrator>') + len('</integrator>')
    (tmp_path / "aov_crop.xml").write_text(text[:start] + aov + text[end:])
    sc = hostlib.HostScene(str(tmp_path / "aov_crop.xml"))
    film, img, st = sc.render()
    flat = sc.flatten()
    ref, _ = oracle.scene(flat).render_aov(flat.params, sc.aov_types())
    assert film.shape == ref.shape == (21, 37, 5 + 1 + 3 + 4)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    # and it is the window of the uncropped film
    plain = hostmirror.write_scene_xml(hostmirror.cbox_meshes(), str(tmp_path), 100, 40, 4, filename="plain.xml")
    text = open(plain).read()
    (tmp_path / "aov_plain.xml").write_text(text[:text.index('<integrator type="path">')] + aov + text[text.index('</integrator>') + len('</integrator>'):])
    full = hostlib.HostScene(str(tmp_path / "aov_plain.xml"))
    whole, _, _ = full.render()
    assert np.array_equal(whole[5:26, 11:48].view(np.uint32), film.view(np.uint32))
    full.close()
    sc.close()
