"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header
declares (no compute calls — there is no GPU here), struct layouts agree, host mirror sanity."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(abi):
    import __graft_entry__ as ge
    ge.build_gpu_library()
    header = open(os.path.join(ROOT, "include", "msk_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(msk_gpu_[a-z_]+)\s*\(", header)))
    assert declared == sorted(abi.EXPORTS)
    lib = C.CDLL(abi.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    # a HIP fat binary for gfx950 is embedded (this is the hand-written kernel build, not a stub)
    blob = open(abi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_shade_gen" in blob and b"k_trace" in blob


def test_struct_layouts_match_the_header(abi, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "msk_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(msk_mesh_desc),sizeof(msk_bsdf_desc),sizeof(msk_emitter_desc),sizeof(msk_camera_desc),'
                   'sizeof(msk_film_desc),sizeof(msk_scene_desc),sizeof(msk_render_params),sizeof(msk_stats),'
                   'offsetof(msk_scene_desc,camera),offsetof(msk_render_params,rng_mode),sizeof(msk_texture_desc),'
                   'offsetof(msk_bsdf_desc,reflectance_texture),offsetof(msk_bsdf_desc,reflectance_scale),offsetof(msk_scene_desc,textures),'
                   'sizeof(msk_spectrum_desc),sizeof(msk_regular_spectrum_desc),offsetof(msk_bsdf_desc,reflectance_regular),'
                   'offsetof(msk_emitter_desc,radiance_regular),offsetof(msk_scene_desc,regular_spectra),offsetof(msk_scene_desc,regular_values),'
                   'offsetof(msk_stats,bytes_shade));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.MeshDesc), C.sizeof(abi.BsdfDesc), C.sizeof(abi.EmitterDesc), C.sizeof(abi.CameraDesc),
            C.sizeof(abi.FilmDesc), C.sizeof(abi.SceneDesc), C.sizeof(abi.RenderParams), C.sizeof(abi.Stats),
            abi.SceneDesc.camera.offset, abi.RenderParams.rng_mode.offset, C.sizeof(abi.TextureDesc),
            abi.BsdfDesc.reflectance_texture.offset, abi.BsdfDesc.reflectance_scale.offset, abi.SceneDesc.textures.offset,
            C.sizeof(abi.SpectrumDesc), C.sizeof(abi.RegularSpectrumDesc), abi.BsdfDesc.reflectance_regular.offset,
            abi.EmitterDesc.radiance_regular.offset, abi.SceneDesc.regular_spectra.offset, abi.SceneDesc.regular_values.offset,
            abi.Stats.bytes_shade.offset]
    assert got == want


def test_missing_library_fails_loudly(abi, tmp_path):
    with pytest.raises(FileNotFoundError) as e:
        abi.load_library(str(tmp_path / "libmsk_gpu.so"))
    assert "no CPU fallback" in str(e.value).replace("NO", "no") or "fallback" in str(e.value)


def test_group_context_argument_checks(abi):
    """msk_gpu_init(ids, n): n < 1 and n > 8 are refused before any device is touched (no GPU needed)."""
    with pytest.raises(abi.MskError) as e:
        abi.Context([0] * 9)
    assert e.value.code == abi.MSK_ERR_INVALID_ARG and "at most 8 devices" in str(e.value)
    with pytest.raises(abi.MskError) as e:
        abi.Context([])
    assert e.value.code == abi.MSK_ERR_INVALID_ARG


def test_flatten_cbox(hostmirror, golden_lookup):
    fs = hostmirror.cbox_scene(512, 512, coeff_lookup=golden_lookup)
    d = fs.desc
    assert (d.n_meshes, d.n_bsdfs, d.n_emitters, d.n_faces, d.n_vertices) == (8, 8, 1, 32, 64)
    assert d.meshes[0].emitter_id == 0 and d.emitters[0].mesh_id == 0
    assert all(d.meshes[i].emitter_id == -1 for i in range(1, 8))
    # srgb_d65.cpp:18-26 + d65.cpp:33-34: scale = 2*max(40) = 80, m_scale = 80 * (1/10568)
    assert d.emitters[0].d65_scale == np.float32(80.0) * (np.float32(1.0) / np.float32(10568.0))
    # quad split (obj.cpp:109-119): (v0,v1,v2) and (v3,v0,v2)
    assert fs.faces[:2].tolist() == [[0, 1, 2], [3, 0, 2]]
    assert d.film.filter_radius == 2.0 and d.film.filter_lut[32] == 0.0


def _sweep_rows(golden):
    import struct
    unhex = lambda h: struct.unpack(">f", bytes.fromhex(h))[0]
    rows = [([unhex(h) for h in r[:3]], [unhex(h) for h in r[3:]]) for r in golden["sweep"]["rows"]]
    rows += [([float(np.float32(x)) for x in v["rgb"]], [unhex(h) for h in v["coeff_hex"]]) for v in golden["triplets"].values()]
    return rows


def test_product_srgb_model_fetch_is_the_references(golden):
    """srgb_model_fetch of the product (python mirror AND C++ host library) against 652 colours fetched by the reference's own
    rgb2spec_fetch from the table the reference's own optimiser wrote (tests/golden/make_golden.py): bit-identical
    coefficients, hence max |dS(lambda)| = 0 over 360-830 nm (the bar asked for was 2e-7)."""
    import importlib
    r2s = importlib.import_module("misaki-render_amd.rgb2spec")
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    lam = np.linspace(360.0, 830.0, 471)
    worst = 0.0
    for rgb, want in _sweep_rows(golden):
        got_py = r2s.srgb_model_fetch(rgb)
        got_cc = hostlib.srgb_model_fetch(rgb)
        worst = max(worst, float(np.abs(r2s.eval_spectrum(got_py, lam) - r2s.eval_spectrum(want, lam)).max()),
                    float(np.abs(r2s.eval_spectrum(got_cc, lam) - r2s.eval_spectrum(want, lam)).max()))
        assert np.array_equal(np.float32(got_py).view(np.uint32), np.float32(want).view(np.uint32)), (rgb, got_py, want)
        assert np.array_equal(np.float32(got_cc).view(np.uint32), np.float32(want).view(np.uint32)), (rgb, got_cc, want)
    assert worst <= 2e-7
    # white is not S = 1 in the reference's table, grey 0.5 is not exactly flat; black is this product's one deviation
    assert 0.9 < r2s.eval_spectrum(r2s.srgb_model_fetch((1, 1, 1)), [550.0])[0] < 0.99
    assert r2s.srgb_model_fetch((0.5, 0.5, 0.5)) != (0.0, 0.0, 0.0)
    assert r2s.srgb_model_fetch((0, 0, 0)) == (0.0, 0.0, -np.inf) == hostlib.srgb_model_fetch((0, 0, 0))


def test_product_table_is_the_references_table(golden, tmp_path):
    """The table srgb_model_fetch reads was computed by the product's own optimiser (host/src/rgb2spec_table.cpp, run by
    `make -C misaki-render_amd/host`): byte for byte the file the reference's rgb2spec_opt wrote in the same image.  A small
    table built here through the C API reproduces the corresponding nodes of the big one's construction (determinism, threads)."""
    import hashlib
    import importlib
    r2s = importlib.import_module("misaki-render_amd.rgb2spec")
    hostlib = importlib.import_module("misaki-render_amd.hostlib")
    assert os.path.basename(r2s.TABLE_PATH) == "srgb.coeff" and "oracle" not in r2s.TABLE_PATH and "reference" not in r2s.TABLE_PATH
    blob = open(r2s.TABLE_PATH, "rb").read()
    assert len(blob) == golden["sweep"]["table_bytes"]
    assert hashlib.sha256(blob).hexdigest() == golden["sweep"]["table_sha256"]
    assert hostlib.srgb_model_source() == r2s.TABLE_PATH
    a, b = tmp_path / "a.coeff", tmp_path / "b.coeff"
    hostlib.rgb2spec_build(10, a, threads=1)
    hostlib.rgb2spec_build(10, b, threads=5)
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(hostlib.HostError) as e:          # too coarse a grid for the warm start: fails loudly, as the reference's tool does
        hostlib.rgb2spec_build(6, tmp_path / "c.coeff")
    assert "singular Jacobian" in str(e.value)
    res, scale, data = r2s.read_table(a)
    assert res == 10 and scale[0] == 0 and scale[-1] == 1 and np.isfinite(data).all()
    # round trip through the model: the spectrum of a table node integrates back to the node's colour
    lam = np.linspace(360.0, 830.0, 941)
    cie = np.array(golden["spectral"]["cie1931_xyz"]).reshape(3, 95)
    d65 = np.array(golden["spectral"]["d65"])
    grid = np.linspace(360.0, 830.0, 95)
    cmf = np.stack([np.interp(lam, grid, c) for c in cie]); ill = np.interp(lam, grid, d65)
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    for l, k, j, i in ((0, 9, 2, 3), (1, 6, 0, 9), (2, 7, 8, 1)):
        rgb = np.zeros(3); bb = float(scale[k])
        rgb[l], rgb[(l + 1) % 3], rgb[(l + 2) % 3] = bb, bb * i / 9.0, bb * j / 9.0
        s = r2s.eval_spectrum(tuple(float(v) for v in data[l, k, j, i]), lam)
        back = m @ ((cmf * ill * s).sum(1) / (cmf[1] * ill).sum())
        assert np.allclose(back, rgb, atol=2e-3), (rgb, back)


def test_develop_matches_hdrfilm_formula(hostmirror):
    film = np.zeros((2, 2, 5), np.float32)
    film[0, 0] = [2, 4, 6, 2, 2]
    img = hostmirror.develop(film)
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    assert np.allclose(img[0, 0, :3], m @ np.array([1, 2, 3.0]), rtol=1e-6) and img[0, 0, 3] == 1.0
    assert np.all(img[1, 1] == 0)       # weight 0 -> 0 (hdrfilm.cpp:72)
