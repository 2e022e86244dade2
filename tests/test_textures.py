"""Textured reflectance: the `checkerboard` texture (textures/checkerboard.cpp:10-33) on the diffuse BSDF's reflectance,
as the reference's own "path" scenes use it for the floor (results/Figure_2_RoughConductor/roughconductor.xml:33-42,
results/Figure_3_RoughDielectric/roughdielectric.xml:31-40).  Oracle pinned by a known-answer table and by radiometry;
GPU compared with the oracle bit for bit."""
import importlib

import numpy as np
import pytest

C0, C1 = (0.725, 0.71, 0.68), (0.325, 0.31, 0.25)              # the reference scenes' colours


def checker_floor_scene(hostmirror, w, h, texture, texcoords=True, twosided=False, lookup=None, extra=()):
    """Cornell box whose floor carries the texture; with texcoords the floor's uv spans [0,1]^2, without them si.uv are the
    hit's barycentrics (mesh.cpp:66)."""
    meshes = hostmirror.cbox_meshes()
    floor = next(m for m in meshes if m.name == "cbox_floor")
    floor.bsdf = {"type": "diffuse", "texture": texture, "twosided": twosided}
    if texcoords:
        floor.texcoords = [((0, 0), (1, 0), (1, 1), (0, 1)) for _ in floor.faces]
    return hostmirror.flatten(meshes + list(extra), w, h, coeff_lookup=lookup)


def test_checkerboard_known_answers(oracle, abi):
    """checkerboard.cpp:24-33 restated independently in numpy fp32: uv' = M3 * (u, v, 1), frac, (u > .5) == (v > .5)."""
    t = abi.TextureDesc()
    t.type = abi.MSK_TEXTURE_CHECKERBOARD
    # hand-checked corners for to_uv = scale(10, 10): cells of 0.05
    t.to_uv[:] = [10, 0, 0, 0, 10, 0]
    table = {(0.01, 0.01): 0, (0.06, 0.01): 1, (0.06, 0.06): 0, (0.01, 0.06): 1, (0.11, 0.01): 0,
             (0.05, 0.01): 0,        # u' = .5 is not > .5
             (-0.01, 0.01): 1,       # frac(-0.1) = 0.9: floor, not truncation
             (-0.01, -0.01): 0, (0.96, 0.99): 0, (1.01, 0.01): 0}
    for (u, v), want in table.items():
        assert oracle.checkerboard(t, u, v) == want, (u, v)
    rng = np.random.RandomState(5)
    for trial in range(20):
        m = rng.uniform(-8, 8, 6).astype(np.float32)
        t.to_uv[:] = m.tolist()
        uv = rng.uniform(-2, 3, (200, 2)).astype(np.float32)
        x = m[0] * uv[:, 0] + (m[1] * uv[:, 1] + m[2])           # fp32 throughout, Eigen's a0 + (a1 + a2)
        y = m[3] * uv[:, 0] + (m[4] * uv[:, 1] + m[5])
        fu, fv = x - np.floor(x), y - np.floor(y)
        want = np.where((fu > np.float32(.5)) == (fv > np.float32(.5)), 0, 1)
        got = np.array([oracle.checkerboard(t, float(a), float(b)) for a, b in uv])
        assert np.array_equal(got, want), trial


def test_descriptor_layout(hostmirror, abi):
    import ctypes
    assert ctypes.sizeof(abi.TextureDesc) == 64 and ctypes.sizeof(abi.BsdfDesc) == 132
    flat = checker_floor_scene(hostmirror, 16, 16, {"type": "checkerboard", "color0": C0, "color1": C1, "scale": (10, 10)})
    d = flat.desc
    assert d.n_textures == 1 and d.textures[0].type == abi.MSK_TEXTURE_CHECKERBOARD
    assert list(d.textures[0].to_uv) == [10, 0, 0, 0, 10, 0]
    tex_users = [i for i in range(d.n_bsdfs) if d.bsdfs[i].reflectance_texture]
    assert len(tex_users) == 1 and d.bsdfs[tex_users[0]].reflectance_texture == 1 and d.bsdfs[tex_users[0]].back_bsdf == -1
    floor = [i for i in range(d.n_meshes) if d.meshes[i].bsdf_id == tex_users[0]]
    assert len(floor) == 1 and d.meshes[floor[0]].has_texcoords == 1
    # the z column of the 4x4 is the uv offset (Transform4f::extract keeps the top-left 3x3, transform.h:142-148)
    m = np.arange(16, dtype=np.float32)
    flat = checker_floor_scene(hostmirror, 16, 16, {"type": "checkerboard", "color0": C0, "color1": C1, "matrix": m.tolist()})
    assert list(flat.desc.textures[0].to_uv) == [0, 1, 2, 4, 5, 6]


def test_oracle_textured_floor_radiometry(oracle, hostmirror, abi):
    """(1) A checkerboard of two equal colours is that colour: films identical to the untextured scene's.
    (2) The pattern is visible: floor pixels split into two brightness populations at the cell positions."""
    w = h = 64
    prm = abi.render_params(32, seed=2)
    plain = hostmirror.cbox_meshes()
    next(m for m in plain if m.name == "cbox_floor").reflectance = C0
    base_sc = oracle.scene(hostmirror.flatten(plain, w, h))
    base, _ = base_sc.render(prm, threads=8)
    base_sc.close()
    same = oracle.scene(checker_floor_scene(hostmirror, w, h, {"type": "checkerboard", "color0": C0, "color1": C0, "scale": (3, 5)},
                                            texcoords=False))      # (texcoords would turn the tangent frame, mesh.cpp:73-79)
    film, _ = same.render(prm, threads=8)
    same.close()
    assert np.array_equal(film, base)
    sc = oracle.scene(checker_floor_scene(hostmirror, w, h, {"type": "checkerboard", "color0": (0.9, 0.9, 0.9), "color1": (0.02, 0.02, 0.02),
                                                               "scale": (2, 2)}))
    film, _ = sc.render(abi.render_params(64, seed=2), threads=8)
    sc.close()
    img = hostmirror.develop(film)[..., 1]
    # scale 2 on uv in [0,1]^2: four quadrants.  The floor's u runs along -x (552.8 -> 0) and pixel x = 0 looks towards +x
    # (SURVEY §8c); the loader stores 1 - v (obj.cpp:96-97), so the front edge (z = 0, vt v = 0) has v = 1.  In the floor's front
    # row (image row 54 of 64) the left part therefore shows frac(u') < .5 with frac(v') > .5: color1, dark, and the part right of
    # the middle color0.
    left, right = img[54, 12:21].mean(), img[54, 24:30].mean()
    assert right > 4 * left, (left, right)


def test_xml_round_trip_through_the_host_library(hostmirror, tmp_path, abi):
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    mat = [3, 0.5, 0.25, 9, -0.5, 4, 0.125, 9, 9, 9, 9, 9, 0, 0, 0, 1]
    for tex, two in (({"type": "checkerboard", "color0": C0, "color1": C1, "scale": (10, 10)}, True),
                     ({"type": "checkerboard", "color0": C1, "color1": C0, "matrix": mat}, False)):
        meshes = hostmirror.cbox_meshes()[:3]
        floor = next(m for m in meshes if m.name == "cbox_floor")
        floor.bsdf = {"type": "diffuse", "texture": tex, "twosided": two}
        floor.texcoords = [((0, 0), (1, 0), (1, 1), (0, 1)) for _ in floor.faces]
        xml = hostmirror.write_scene_xml(meshes, str(tmp_path), 16, 16, 1)
        hs = hostlib.HostScene(xml)
        d = hs.flatten().desc
        r = hostmirror.flatten(meshes, 16, 16).desc
        assert d.n_textures == r.n_textures == 1
        a, b = d.textures[0], r.textures[0]
        assert a.type == b.type and list(a.to_uv) == list(b.to_uv)
        assert np.allclose(a.color0[:], b.color0[:], rtol=2e-4, atol=2e-6) and np.allclose(a.color1[:], b.color1[:], rtol=2e-4, atol=2e-6)
        for i in range(3):
            assert d.meshes[i].has_texcoords == r.meshes[i].has_texcoords
            ba, bb = d.bsdfs[d.meshes[i].bsdf_id], r.bsdfs[r.meshes[i].bsdf_id]
            assert ba.reflectance_texture == bb.reflectance_texture and (ba.back_bsdf >= 0) == (bb.back_bsdf >= 0)
        nv = d.n_vertices
        va = np.ctypeslib.as_array(d.vertices, (nv * 8,)).reshape(nv, 8)
        vb = np.ctypeslib.as_array(r.vertices, (nv * 8,)).reshape(nv, 8)
        assert np.array_equal(va[:, 6:], vb[:, 6:]) and np.array_equal(va[:, :3], vb[:, :3])
    # defaults of the plugin (checkerboard.cpp:11-13): color0 .4, color1 .2, identity to_uv
    text = open(xml).read()
    start, end = text.index('<texture name="reflectance"'), text.index('</texture>') + len('</texture>')
    (tmp_path / "dflt.xml").write_text(text[:start] + '<texture name="reflectance" type="checkerboard"/>' + text[end:])
    t = hostlib.HostScene(str(tmp_path / "dflt.xml")).flatten().desc.textures[0]
    assert list(t.to_uv) == [1, 0, 0, 0, 1, 0] and not np.allclose(t.color0[:], t.color1[:])
    # a texture the back end cannot flatten is an error of render()/flatten, not a silent constant
    (tmp_path / "nested.xml").write_text(text[:start] + '<texture name="reflectance" type="checkerboard"><texture name="color0" type="checkerboard"/></texture>' + text[end:])
    with pytest.raises(hostlib.HostError) as e:
        hostlib.HostScene(str(tmp_path / "nested.xml")).flatten()
    assert "cannot be evaluated by the GPU path integrator" in str(e.value)


@pytest.mark.gpu
def test_gpu_matches_oracle_on_textured_floors(gpu_ctx, oracle, hostmirror, golden_lookup, abi):
    mat = [7, 1.5, 0.3, 0, -2, 6, -0.2, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    ball = hostmirror.blob_mesh("ball", (370, 90, 170), 80, 20, 20, hostmirror.WHITE, seed=2)
    ball.bsdf = {"type": "roughconductor", "alpha": 0.15, "eta": (0.2, 0.92, 1.1), "k": (3.9, 2.45, 2.14), "twosided": True}
    cases = [dict(texture={"type": "checkerboard", "color0": C0, "color1": C1, "scale": (10, 10)}, texcoords=True, twosided=True),
             dict(texture={"type": "checkerboard", "color0": C0, "color1": C1, "scale": (4, 4)}, texcoords=False),
             dict(texture={"type": "checkerboard", "color0": C1, "color1": C0, "matrix": mat}, texcoords=True, extra=[ball])]
    for kw in cases:
        flat = checker_floor_scene(hostmirror, 96, 96, **kw)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        prm = abi.render_params(spp=16, seed=11)
        rng = np.random.RandomState(4)
        pixels = np.concatenate([rng.randint(0, 96, (40, 2)), np.c_[rng.randint(16, 80, 40), rng.randint(76, 84, 40)]]).astype(np.int32)   # + the floor
        gx, gp = g.sample_pixels(prm, pixels)
        ox, op = o.sample_pixels(prm, pixels)
        assert np.array_equal(gp, op)
        bad = (gx.view(np.uint32) != ox.view(np.uint32)).any(-1)
        assert not bad.any(), (int(bad.sum()), gx[bad][:3], ox[bad][:3])
        film, st = g.render(abi.render_params(spp=8, seed=5))
        ref, rst = o.render(abi.render_params(spp=8, seed=5), threads=8)
        assert np.array_equal(film, ref), float(np.abs(film - ref).max())
        assert st.samples == rst.samples and 0.97 * rst.segments <= st.segments <= rst.segments   # zero-throughput rays are not traced
        g.close(); o.close()


@pytest.mark.gpu
def test_gpu_rejects_bad_texture_descriptors(gpu_ctx, hostmirror, abi):
    flat = checker_floor_scene(hostmirror, 16, 16, {"type": "checkerboard", "color0": C0, "color1": C1, "scale": (10, 10)})
    flat.desc.textures[0].type = 7
    with pytest.raises(abi.MskError) as e:
        abi.Scene(gpu_ctx, flat)
    assert "texture 0: type 7" in str(e.value)
    flat.desc.textures[0].type = abi.MSK_TEXTURE_CHECKERBOARD
    flat.desc.n_textures = 0
    with pytest.raises(abi.MskError) as e:
        abi.Scene(gpu_ctx, flat)
    assert "reflectance_texture 1 out of range" in str(e.value)


FIGURE_STYLE_XML = """<?xml version="1.0" encoding="utf-8"?>
<scene>
    <integrator type="path"/>
    <sensor type="perspective">
        <float name="fov" value="35"/>
        <transform name="to_world">
            <matrix value="-1 0 0 278  0 1 0 273  0 0 -1 1400  0 0 0 1"/>
        </transform>
        <sampler type="independent"><integer name="sample_count" value="4"/></sampler>
        <film type="rgbfilm"><integer name="width" value="64"/><integer name="height" value="36"/></film>
    </sensor>
    <bsdf type="roughdielectric" id="Material">
        <float name="alpha" value="0.1"/>
        <string name="distribution" value="ggx"/>
        <rgb name="specular_reflectance" value="1, 1, 1"/>
        <rgb name="eta" value="0.2, 0.92, 1.1"/>
    </bsdf>
    <bsdf type="twosided" id="Stand"><bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.2, 0.2"/></bsdf></bsdf>
    <bsdf type="twosided" id="Floor">
        <bsdf type="diffuse">
            <texture name="reflectance" type="checkerboard">
                <rgb name="color1" value="0.325, 0.31, 0.25"/>
                <rgb name="color0" value="0.725, 0.71, 0.68"/>
                <transform name="to_uv"><scale x="10" y="10"/></transform>
            </texture>
        </bsdf>
    </bsdf>
    <shape type="obj">
        <transform name="to_world"><matrix value="2 0 0 -278  0 1 0 0  0 0 2 -280  0 0 0 1"/></transform>
        <string name="filename" value="meshes/cbox_floor.obj"/>
        <ref id="Floor"/>
    </shape>
    <shape type="obj"><string name="filename" value="meshes/cbox_smallbox.obj"/><ref id="Material"/></shape>
    <shape type="obj"><string name="filename" value="meshes/cbox_largebox.obj"/><ref id="Stand"/></shape>
    <emitter type="constant">
        <transform name="to_world"><matrix value="-0.92 0 0.39 0  0 1 0 0  -0.39 0 -0.92 1.17  0 0 0 1"/></transform>
        <string name="filename" value="textures/envmap.hdr"/>
    </emitter>
</scene>
"""


def figure_style_scene(hostmirror, tmp_path):
    """A scene file shaped like the reference's Figure 2 / 3 `"path"` scenes (results/Figure_3_RoughDielectric/
    roughdielectric.xml): named top-level BSDFs referenced by <ref>, <matrix> transforms on the sensor and a shape, a
    checkerboard floor under `twosided`, rgbfilm, a `constant` emitter carrying properties its plugin never reads."""
    meshes = hostmirror.cbox_meshes()
    floor = next(m for m in meshes if m.name == "cbox_floor")
    floor.texcoords = [((0, 0), (1, 0), (1, 1), (0, 1)) for _ in floor.faces]
    hostmirror.write_scene_xml(meshes, str(tmp_path), 16, 16, 1)            # writes meshes/*.obj
    path = tmp_path / "figure.xml"
    path.write_text(FIGURE_STYLE_XML)
    return str(path)


def test_figure_style_scene_loads_and_flattens(hostmirror, oracle, tmp_path, abi):
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    hs = hostlib.HostScene(figure_style_scene(hostmirror, tmp_path))
    flat = hs.flatten()
    d = flat.desc
    assert (d.film.width, d.film.height, d.n_meshes, d.n_emitters, d.n_textures) == (64, 36, 3, 1, 1)
    assert d.emitters[0].type == abi.MSK_EMITTER_CONSTANT and d.meshes[0].has_texcoords == 1
    fb = d.bsdfs[d.meshes[0].bsdf_id]
    assert fb.type == abi.MSK_BSDF_DIFFUSE and fb.reflectance_texture == 1 and fb.back_bsdf == d.meshes[0].bsdf_id
    assert d.bsdfs[d.meshes[1].bsdf_id].type == abi.MSK_BSDF_ROUGHDIELECTRIC
    assert list(d.textures[0].to_uv) == [10, 0, 0, 0, 10, 0]
    nv = d.meshes[0].vertex_count
    v = np.ctypeslib.as_array(d.vertices, (d.n_vertices * 8,)).reshape(-1, 8)[:nv]
    assert np.allclose(v[0, :3], [2 * 552.8 - 278, 0, -280])                # the shape's <matrix> was applied at load
    osc = oracle.scene(flat)
    film, st = osc.render(flat.params, threads=4)
    assert np.isfinite(film).all() and film[..., :3].min() > 0              # the environment lights every pixel
    osc.close(); hs.close()


@pytest.mark.gpu
def test_figure_style_scene_renders_like_the_oracle(hostmirror, oracle, tmp_path, abi):
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    hs = hostlib.HostScene(figure_style_scene(hostmirror, tmp_path))
    flat = hs.flatten()
    got, rgba, st = hs.render()                                             # the "path" plugin: flatten -> C ABI -> film
    osc = oracle.scene(flat)
    film, _ = osc.render(flat.params, threads=8)
    osc.close(); hs.close()
    assert got.shape == (36, 64, 5) and st.samples == 64 * 36 * 4
    assert np.array_equal(got.view(np.uint32), film.view(np.uint32)), float(np.abs(got - film).max())
    assert np.allclose(rgba, hostmirror.develop(film), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
# `uniform` spectrum (spectra/uniform.cpp): what <spectrum name="reflectance" value="c"/> creates outside an emitter
# (xml.cpp:285-292) — the value c at every wavelength, not the sRGB upsampling of (c, c, c)
# ---------------------------------------------------------------------------------------------------------------------
def uniform_scene(hostmirror, w, h, c=0.37):
    meshes = hostmirror.cbox_meshes()
    for m in meshes:
        if m.name in ("cbox_floor", "cbox_back"):
            m.reflectance = c
    return meshes, hostmirror.flatten(meshes, w, h)


def test_uniform_reflectance_descriptor_and_eval(oracle, hostmirror, abi):
    meshes, flat = uniform_scene(hostmirror, 16, 16)
    d = flat.desc
    floor = next(i for i, m in enumerate(meshes) if m.name == "cbox_floor")
    b = d.bsdfs[d.meshes[floor].bsdf_id]
    assert list(b.reflectance[:2]) == [0, 0] and np.isinf(b.reflectance[2]) and b.reflectance_scale == np.float32(0.37)
    other = d.bsdfs[d.meshes[floor + 1].bsdf_id]
    assert other.reflectance_scale == 1.0
    # diffuse.cpp:44: f = reflectance / pi * cos(theta_o), the same at every wavelength
    bs = [d.bsdfs[i] for i in range(d.n_bsdfs)]
    wo = np.array([0.3, -0.2, np.sqrt(1 - 0.13)], np.float32)
    val, pdf = oracle.bsdf_eval(bs, d.meshes[floor].bsdf_id, (0, 0, 1), wo, wl=(380, 500, 620, 800))
    want = np.float32(0.37) * np.float32(1 / np.pi) * wo[2]
    assert np.all(val == val[0]) and np.isclose(val[0], want, rtol=3e-7) and np.isclose(pdf, wo[2] / np.pi, rtol=3e-7)


def test_uniform_reflectance_xml_round_trip(hostmirror, tmp_path, abi):
    import __graft_entry__ as ge
    ge.build_gpu_library()
    ge.build_host_library()
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    meshes, ref = uniform_scene(hostmirror, 16, 16)
    xml = hostmirror.write_scene_xml(meshes, str(tmp_path), 16, 16, 1)
    assert '<spectrum name="reflectance" value="0.37"/>' in open(xml).read()
    d, r = hostlib.HostScene(xml).flatten().desc, ref.desc
    for i in range(d.n_meshes):
        a, b = d.bsdfs[d.meshes[i].bsdf_id], r.bsdfs[r.meshes[i].bsdf_id]
        assert a.reflectance_scale == b.reflectance_scale and np.isinf(a.reflectance[2]) == np.isinf(b.reflectance[2])


@pytest.mark.gpu
def test_gpu_matches_oracle_with_uniform_reflectances(gpu_ctx, oracle, hostmirror, abi):
    for c in (0.37, 1.0, 0.0):
        _, flat = uniform_scene(hostmirror, 64, 64, c)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        prm = abi.render_params(spp=8, seed=3)
        film, st = g.render(prm)
        ref, rst = o.render(prm, threads=8)
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), c
        g.close(); o.close()
    _, flat = uniform_scene(hostmirror, 16, 16)
    flat.desc.bsdfs[0].reflectance_scale = float("nan")
    with pytest.raises(abi.MskError) as e:
        abi.Scene(gpu_ctx, flat)
    assert "reflectance_scale" in str(e.value)
