"""Known-answer tests that PIN the CPU oracle (CPU only, no GPU).

The reference ships no tests, so the vectors here are the ones SURVEY §8(c) captured
from the reference's own formulas / from the one leaf of the reference that builds here
(ext/rgb2spec -> oracle/_ref), plus hand-derived closed-form values.
"""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hexf(h):
    return struct.unpack(">f", bytes.fromhex(h))[0]


# ----------------------------------------------------------------------------- a16 PCG32
def test_pcg32_golden(oracle):
    u, f, si = oracle.pcg32(0x853c49e6748fea9b, 0xda3e39cb94b95bdb, 8)
    assert int(si[0]) == 0xea1b84321029ee21 and int(si[1]) == 0xb47c73972972b7b7
    assert [int(x) for x in u] == [0x1bbeb4f2, 0xe82e89e9, 0x681cfdeb, 0xe00fa2ec, 0xb1e1a434, 0xbe56068d,
                                   0x2add8c94, 0x9f1b63f5]
    want = [0.10837865, 0.90696001, 0.40669227, 0.87523854, 0.69484925, 0.74350011, 0.16744304, 0.62151158]
    assert np.allclose(f, want, rtol=0, atol=5e-9)
    # float = (u >> 9 | 0x3f800000) - 1   (mathutils.h:111-121)
    ref = ((u >> 9) | 0x3f800000).view(np.float32) - np.float32(1)
    assert np.array_equal(f, ref)


def test_constants(oracle):
    eps, ray, shadow = oracle.constants()
    assert eps == np.float32(5.9604645e-08) and ray == np.float32(8.940697e-05) and shadow == np.float32(8.940697e-04)


def test_counter_rng_is_stateless_and_uniform(oracle):
    a = oracle.counter_pair(7, 1234, 5, 3)
    assert np.array_equal(a, oracle.counter_pair(7, 1234, 5, 3))
    assert not np.array_equal(a, oracle.counter_pair(8, 1234, 5, 3))
    assert not np.array_equal(a, oracle.counter_pair(7, 1235, 5, 3))
    assert not np.array_equal(a, oracle.counter_pair(7, 1234, 6, 3))
    assert not np.array_equal(a, oracle.counter_pair(7, 1234, 5, 4))
    v = np.array([oracle.counter_pair(0, p, s, 0) for p in range(64) for s in range(64)])
    assert 0.0 <= v.min() and v.max() < 1.0
    assert abs(v.mean() - 0.5) < 0.01 and abs(v.var() - 1 / 12) < 0.005
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 0.03
    # python restatement of the definition
    M = (1 << 64) - 1

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
        return z ^ (z >> 31)
    key = mix((((1234 << 32) | 5) + 0x9e3779b97f4a7c15 * (7 + 1)) & M)
    r = mix((key + 0x9e3779b97f4a7c15 * (3 + 1)) & M)
    f = lambda u32: np.array([(u32 >> 9) | 0x3f800000], np.uint32).view(np.float32)[0] - np.float32(1)
    assert a[0] == f(r >> 32) and a[1] == f(r & 0xffffffff)


# ----------------------------------------------------------------------------- a14 filter
def test_gaussian_lut_golden(oracle, hostmirror):
    radius, lut, scale, border = oracle.gaussian_filter(0.5)
    assert radius == 2.0 and border == 2 and scale == 16.0
    assert np.allclose(lut[:4], [0.76056546, 0.75464475, 0.7371574, 0.70890766], rtol=2e-7)
    assert np.isclose(lut[31], 1.6229642e-4, rtol=2e-6) and lut[32] == 0.0
    r2, lut2 = hostmirror.gaussian_filter(0.5)          # the product's host mirror
    assert r2 == radius and np.allclose(lut2, lut, rtol=3e-7, atol=1e-9)


# ----------------------------------------------------------------------------- a4 wavelengths
def test_wavelength_sampling_golden(oracle):
    wl, w = oracle.sample_wavelength(0.5)
    assert np.allclose(wl, [545.903, 616.8562, 830.0, 479.15405], rtol=2e-7)
    assert np.allclose(w, [254.64272, 344.81903, 4379.7993, 302.17682], rtol=3e-6)
    # shift rule value <= 1 keeps exactly 1.0 un-wrapped (mathutils.h:174-176): u = 0.5 + 0.5 = 1 -> 830 nm
    assert wl[2] == np.float32(830.0)
    oracle.set_libm(1)
    wl2, w2 = oracle.sample_wavelength(0.5)
    oracle.set_libm(0)
    assert np.allclose(wl, wl2, rtol=3e-7) and np.allclose(w, w2, rtol=3e-6)


def test_det_math_within_one_ulp_of_libm(oracle):
    rng = np.random.RandomState(1)
    xs = np.concatenate([rng.uniform(-0.97, 0.86, 2000), [0.0, 0.5, -0.5]]).astype(np.float32)
    worst = np.zeros(4)
    for x in xs:
        d = oracle.det_math(float(x)).astype(np.float64)
        xd = float(x)
        ref = np.array([np.sin(xd), np.cos(xd), np.arctanh(xd), np.cosh(xd)])
        ulp = np.abs(np.spacing(ref.astype(np.float32))).astype(np.float64)
        worst = np.maximum(worst, np.abs(d - ref) / ulp)
    assert worst.max() <= 0.5 + 1e-6, worst     # correctly rounded
    # cosh over the whole range the wavelength weights use (|0.0072 (lambda - 538)| <= 2.11) and beyond the series' 2.5
    for x in np.concatenate([rng.uniform(-2.2, 2.2, 1500), rng.uniform(-9, 9, 300), [2.5, -2.5, 2.5000002]]).astype(np.float32):
        got, ref = float(oracle.det_math(float(x))[3]), np.cosh(float(x))
        assert abs(got - ref) <= (0.5 + 1e-6) * abs(np.spacing(np.float32(ref))), x
    # atanh towards the ends of (-1, 1): exponent bookkeeping of the single-division form
    for x in np.concatenate([rng.uniform(-0.999999, 0.999999, 1500), [0.99999994, -0.99999994, 5.9604645e-8, -3e-7]]).astype(np.float32):
        got, ref = float(oracle.det_math(float(x))[2]), np.arctanh(float(x))
        assert abs(got - ref) <= (0.5 + 1e-6) * abs(np.spacing(np.float32(ref))), x
    # angles beyond the first quadrant (quadrant reduction)
    for x in np.linspace(-20, 20, 401, dtype=np.float32):
        d = oracle.det_math(float(np.float32(x) * np.float32(0.999)))
        xd = float(np.float32(x) * np.float32(0.999))
        assert abs(d[0] - np.sin(xd)) <= 6e-8 and abs(d[1] - np.cos(xd)) <= 6e-8


# ----------------------------------------------------------------------------- a12 spectra
def test_rgb2spec_fetch_matches_reference_build(oracle, golden):
    coeff = os.path.join(ROOT, "oracle", "_ref", "srgb.coeff")
    so = os.path.join(ROOT, "oracle", "_ref", "librgb2spec.so")
    if not os.path.exists(coeff):
        pytest.skip("oracle/_ref not built (make -C oracle ref; needs /root/reference)")
    raw = open(coeff, "rb").read()
    assert raw[:4] == b"SPEC"
    res = struct.unpack("<I", raw[4:8])[0]
    scale = np.frombuffer(raw, np.float32, res, 8).copy()
    data = np.frombuffer(raw, np.float32, 3 * 3 * res ** 3, 8 + 4 * res).copy()
    for v in golden["triplets"].values():
        got = oracle.rgb2spec_fetch(res, scale, data, v["rgb"])
        assert [struct.pack(">f", x).hex() for x in got] == v["coeff_hex"], v
    # and against the reference's compiled rgb2spec_fetch on random colours, bit for bit
    lib = C.CDLL(so)
    lib.rgb2spec_load.restype = C.c_void_p
    lib.rgb2spec_load.argtypes = [C.c_char_p]
    lib.rgb2spec_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    model = lib.rgb2spec_load(coeff.encode())
    rng = np.random.RandomState(3)
    for rgb in rng.uniform(0.001, 1, (500, 3)).astype(np.float32):
        a, o = (C.c_float * 3)(*rgb), (C.c_float * 3)()
        lib.rgb2spec_fetch(model, a, o)
        assert np.array_equal(np.array(o[:], np.float32), oracle.rgb2spec_fetch(res, scale, data, rgb))


def test_srgb_model_eval(oracle, golden):
    wl = np.array([545.903, 616.8562, 830.0, 479.15405], np.float32)
    for name, v in golden["triplets"].items():
        c = np.array([hexf(h) for h in v["coeff_hex"]], np.float32)
        got = oracle.srgb_model_eval(c, wl)
        if np.isinf(c[2]):
            assert np.all(got == (1.0 if c[2] > 0 else 0.0))
            continue
        x = (c[0].astype(np.float64) * wl + c[1]) * wl + c[2]
        ref = 0.5 + 0.5 * x / np.sqrt(1 + x * x)
        assert np.allclose(got, ref, rtol=0, atol=2e-5 + 2e-5 * np.abs(x).max()), name
        assert np.all((got >= 0) & (got <= 1))
    # a grey-ish colour keeps its level; white stays near 0.75-0.9 across the visible range
    white = np.array([hexf(h) for h in golden["triplets"]["white"]["coeff_hex"]], np.float32)
    s = oracle.srgb_model_eval(white, np.array([450, 550, 600, 650], np.float32))
    assert np.all((s > 0.6) & (s < 0.95))


# ----------------------------------------------------------------------------- math core
def test_coordinate_system(oracle):
    s, t = oracle.coordinate_system([0, 0, 1])
    assert np.array_equal(s, [1, 0, 0]) and np.array_equal(t, [0, 1, 0])
    s, t = oracle.coordinate_system([0, 0, -1])
    assert np.array_equal(s, [1, 0, 0]) and np.array_equal(t, [0, -1, 0])
    rng = np.random.RandomState(0)
    for n in rng.normal(size=(200, 3)):
        n = (n / np.linalg.norm(n)).astype(np.float32)
        s, t = oracle.coordinate_system(n)
        m = np.stack([s, t, n]).astype(np.float64)
        assert np.allclose(m @ m.T, np.eye(3), atol=5e-7)
        assert np.allclose(np.cross(s, t), n, atol=5e-7)         # right-handed (s, t, n)


def test_warps(oracle):
    tri, disk, hemi = oracle.warps([0.25, 0.5])
    t = np.sqrt(np.float32(0.75))
    assert tri[0] == np.float32(1) - t and tri[1] == t * np.float32(0.5)
    _, disk, hemi = oracle.warps([0.5, 0.5])
    assert np.array_equal(disk, [0, 0]) and np.array_equal(hemi, [0, 0, 1])
    for u in [(i / 4 + 0.1, j / 4 + 0.05) for i in range(4) for j in range(4)]:
        tri, disk, hemi = oracle.warps(u)
        assert tri[0] >= 0 and tri[1] >= 0 and tri[0] + tri[1] <= 1
        assert disk @ disk <= 1 + 1e-6 and abs(hemi @ hemi - 1) < 1e-6 and hemi[2] >= 0
        # concentric map: radius = max(|2u-1|), angle within the matching octant pair
        x, y = 2 * u[0] - 1, 2 * u[1] - 1
        assert abs(np.sqrt(disk @ disk) - max(abs(x), abs(y))) < 1e-6
    # x*x > y*y branch: phi = pi/4 * y/x
    _, disk, _ = oracle.warps([0.9, 0.6])
    r, phi = 0.8, np.pi / 4 * (0.2 / 0.8)
    assert np.allclose(disk, [r * np.cos(phi), r * np.sin(phi)], atol=1e-6)


# ----------------------------------------------------------------------------- a3 camera
def test_camera_golden(oracle, hostmirror):
    def rays(w, h, pts):
        fs = hostmirror.cbox_scene(w, h, coeff_lookup=lambda rgb: (0.0, 0.0, 0.0))
        sc = oracle.scene(fs)
        out = [sc.camera_ray(0.5, *p)[0] for p in pts]
        sc.close()
        return out
    r = rays(512, 512, [(256, 256), (0, 0), (512, 512)])
    assert np.allclose(r[0][4:7], [0, 0, 1], atol=1e-7) and np.isclose(r[0][3], 10) and np.isclose(r[0][7], 2800)
    assert np.allclose(r[0][:3], [278, 273, -800])
    assert np.allclose(r[1][4:7], [0.384984, 0.384984, 0.838794], atol=2e-6)
    assert np.isclose(r[1][3], 11.9219, rtol=1e-5) and np.isclose(r[1][7], 3338.13, rtol=1e-5)
    assert np.allclose(r[2][4:7], [-0.384984, -0.384984, 0.838794], atol=2e-6)
    r = rays(1920, 1080, [(0, 0)])
    assert np.allclose(r[0][4:7], [0.406106, 0.228435, 0.884814], atol=2e-6) and np.isclose(r[0][3], 11.3018, rtol=1e-5)
    # the product's host mirror builds the same matrices as the oracle's restatement
    cam = hostmirror.CBOX_CAMERA
    a = oracle.perspective_camera(cam["fov"], cam["near"], cam["far"], 1920, 1080, cam["origin"], cam["target"], cam["up"])
    b = hostmirror.perspective_camera(cam["fov"], cam["near"], cam["far"], 1920, 1080, cam["origin"], cam["target"], cam["up"])
    assert np.allclose(a[0], b[0], rtol=1e-6, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-6, atol=1e-9)


# ----------------------------------------------------------------------------- a1 block generator
def test_spiral_block_order(oracle):
    b = oracle.spiral_blocks(256, 256)
    assert len(b) == 64
    assert [tuple(x[:2]) for x in b[:7]] == [(128, 128), (160, 128), (160, 160), (128, 160), (96, 160), (96, 128),
                                             (96, 96)]
    assert len({tuple(x[:2]) for x in b}) == 64 and np.all(b[:, 2:] == 32)
    b = oracle.spiral_blocks(1920, 1080)
    assert len(b) == 60 * 34 and len({tuple(x[:2]) for x in b}) == 2040
    assert tuple(b[0][:2]) == (30 * 32, 17 * 32)
    last_row = b[b[:, 1] == 33 * 32]
    assert len(last_row) == 60 and np.all(last_row[:, 3] == 24)      # 1080 - 33*32 = 24 (imageblock.cpp:206-208)
    cover = np.zeros((1080, 1920), np.int32)
    for ox, oy, sx, sy in b:
        cover[oy:oy + sy, ox:ox + sx] += 1
    assert np.all(cover == 1)
    b = oracle.spiral_blocks(100, 40)        # ragged edges
    assert sorted((int(x[0]), int(x[1]), int(x[2]), int(x[3])) for x in b) == sorted(
        (ox, oy, min(32, 100 - ox), min(32, 40 - oy)) for ox in (0, 32, 64, 96) for oy in (0, 32))


# ----------------------------------------------------------------------------- a14 ImageBlock::put
def test_imageblock_put(oracle, hostmirror):
    fs = hostmirror.cbox_scene(64, 64, coeff_lookup=lambda rgb: (0.0, 0.0, 0.0))
    film = fs.desc.film
    lut = np.array(film.filter_lut[:], np.float32)
    val = np.array([[1, 2, 3, 1, 1]], np.float32)
    # a sample in the middle of pixel (32+5, 32+7) of the block at offset (32,32)
    out = oracle.block_put(film, (32, 32), (32, 32), [[37.5, 39.5]], val)
    assert out.shape == (36, 36, 5)
    # pos' = 37.5 - .5 - (32 - 2) = 7 -> lo = ceil(7-2) = 5, hi = floor(7+2) = 9
    nz = np.argwhere(out[..., 4] != 0)
    assert nz[:, 1].min() == 6 and nz[:, 1].max() == 8      # |x|=2 hits LUT[32] = 0
    wx = np.array([lut[min(int(abs(x - 7.0) * 16), 32)] for x in range(5, 10)], np.float32)
    wy = np.array([lut[min(int(abs(y - 9.0) * 16), 32)] for y in range(7, 12)], np.float32)
    ref = np.outer(wy, wx).astype(np.float32)
    assert np.array_equal(out[7:12, 5:10, 4], ref)
    assert np.array_equal(out[7:12, 5:10, 1], ref * np.float32(2))
    # corner sample: footprint clipped by the bordered tile, nothing written outside
    out = oracle.block_put(film, (0, 0), (32, 32), [[0.1, 0.2]], val)
    nz = np.argwhere(out[..., 4] != 0)
    assert nz.min() == 0 and nz[:, 0].max() <= 4 and nz[:, 1].max() <= 4
    # accumulation is sequential fp32: two puts == sum in order
    a = oracle.block_put(film, (0, 0), (32, 32), [[10.3, 11.7], [10.9, 11.2]], np.array([[1, 2, 3, 1, 1], [4, 5, 6, 1, 1]], np.float32))
    b1 = oracle.block_put(film, (0, 0), (32, 32), [[10.3, 11.7]], val)
    b2 = oracle.block_put(film, (0, 0), (32, 32), [[10.9, 11.2]], np.array([[4, 5, 6, 1, 1]], np.float32))
    assert np.array_equal(a, b1 + b2)


# ----------------------------------------------------------------------------- a10 light tables
def test_luminaire_area_and_cdf(oracle, hostmirror):
    fs = hostmirror.cbox_scene(32, 32, coeff_lookup=lambda rgb: (0.0, 0.0, 0.0))
    sc = oracle.scene(fs)
    area, cdf = sc.mesh_tables(0)
    assert area == 130.0 * 105.0 and np.array_equal(cdf, [0, 0.5, 1.0])
    area, cdf = sc.mesh_tables(6)       # small box: 10 triangles
    assert len(cdf) == 11 and cdf[0] == 0 and cdf[-1] == 1 and np.all(np.diff(cdf) > 0)
    sc.close()


def test_cbox_winding(hostmirror):
    """Walls face inward, blocks outward, the light downward (one-sided BSDF / emitter)."""
    centre = np.array([278, 274, 280.0])
    for m in hostmirror.cbox_meshes():
        v, f = hostmirror.triangulate(m)
        p = v[:, :3].astype(np.float64)
        for tri in f:
            n = np.cross(p[tri[1]] - p[tri[0]], p[tri[2]] - p[tri[0]])
            c = p[tri].mean(0)
            if m.name in ("cbox_smallbox", "cbox_largebox"):
                assert n @ (c - p.mean(0)) > 0, m.name
            elif m.name == "cbox_luminaire":
                assert n[1] < 0 and n[0] == 0 and n[2] == 0
            else:
                assert n @ (centre - c) > 0, m.name
    v, _ = hostmirror.triangulate(hostmirror.cbox_meshes()[0])
    assert np.all(v[:, 1] == np.float32(548.8) + np.float32(-0.5))


# ----------------------------------------------------------------------------- F7 / D2 samplers
def test_pcg_block_and_counter_modes_estimate_the_same_image(oracle, hostmirror, abi):
    """The reference-semantics sampler (one PCG32 stream per 32x32 block, D2) and the counter RNG
    the GPU shares estimate the same image: tile means differ no more than two counter-mode
    renders with different seeds do.  Also: libm vs det_* transcendentals (D7)."""
    fs = hostmirror.cbox_scene(128, 128)
    sc = oracle.scene(fs)
    tiles = lambda f: hostmirror.develop(f)[..., :3].reshape(4, 32, 4, 32, 3).mean((1, 3))
    a = tiles(sc.render(abi.render_params(32, seed=0), 8)[0])
    b = tiles(sc.render(abi.render_params(32, seed=0, rng_mode=abi.MSK_RNG_PCG_BLOCK), 8)[0])
    c = tiles(sc.render(abi.render_params(32, seed=1), 8)[0])
    noise = np.abs(a - c).max()
    assert np.abs(a - b).max() <= 2.5 * noise + 1e-3
    assert np.all(np.abs(a - b) <= 0.03 * (np.abs(a) + 0.3))
    assert abs(a.mean() - b.mean()) < 3e-3
    oracle.set_libm(1)
    d_film, _ = sc.render(abi.render_params(32, seed=0), 8)
    oracle.set_libm(0)
    d = tiles(d_film)
    assert np.all(np.abs(a - d) <= 2e-3 * (np.abs(a) + 0.3))      # last-ulp differences only flip rare decisions
    sc.close()


# ----------------------------------------------------------------------------- regression anchor of the oracle itself
def test_oracle_films_match_the_committed_anchor(oracle, hostmirror, abi):
    """GPU parity is measured against the oracle, so a change that moved both together would go unnoticed: three small
    films of the oracle (diffuse cbox; metal + glass; open box + environment with rr_depth 2) are pinned by SHA-256
    (tests/golden/oracle_film.json, written by tests/golden/make_film_golden.py).  Update the anchor deliberately."""
    import hashlib
    import importlib.util
    here = os.path.join(ROOT, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_film_golden", os.path.join(here, "make_film_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    anchor = json.load(open(os.path.join(here, "oracle_film.json")))
    trip = json.load(open(os.path.join(here, "rgb2spec_triplets.json")))
    table = {tuple(round(c, 6) for c in v["rgb"]): tuple(struct.unpack(">f", bytes.fromhex(h))[0] for h in v["coeff_hex"])
             for v in trip.values()}
    seen = 0
    for name, flat, kw in mod.scenes(hostmirror, lambda rgb: table[tuple(round(c, 6) for c in rgb)]):
        sc = oracle.scene(flat)
        film, st = sc.render(abi.render_params(**kw), threads=3)
        sc.close()
        a = anchor[name]
        assert list(film.shape) == a["shape"] and int(st.samples) == a["samples"]
        for key, hexval in a["pixels_hex"].items():
            y, x = (int(v) for v in key.split(","))
            assert film[y, x].tobytes().hex() == hexval, (name, key)
        assert hashlib.sha256(film.tobytes()).hexdigest() == a["sha256"], name
        seen += 1
    assert seen == 3


def degenerate_scenes(hostmirror):
    """Scenes at the edges of the descriptor: nothing at all, an environment and no geometry, a mesh without faces, an area
    light of zero area, one triangle."""
    light = hostmirror.MeshSpec("l", [((0, 0, 0), (0, 0, 0), (0, 0, 0))], hostmirror.WHITE, radiance=(1, 1, 1))
    tri = hostmirror.MeshSpec("t", [((100, 100, 300), (280, 420, 280), (450, 120, 320))], hostmirror.RED, radiance=(3, 2, 1))
    return {"nothing": ([], None), "environment only": ([], {"radiance": None}),
            "mesh without faces": ([hostmirror.MeshSpec("e", [], hostmirror.WHITE)], {"radiance": (0.2, 0.3, 0.4)}),
            "zero-area light": ([light] + hostmirror.cbox_meshes()[1:4], None),
            "one emitting triangle": ([tri], None)}


def test_oracle_crop_window_is_a_window_of_the_full_film(oracle, abi, hostmirror):
    """film.cpp:12-21, hdrfilm.cpp:37-38, imageblock.cpp:133-173: HDRFilm's storage is the crop window, every block is clipped
    to it in spiral order — so the cropped film is, bit for bit, that window of the full film, for tile shards too."""
    full = oracle.scene(hostmirror.cbox_scene(70, 50))
    for crop in ((11, 5, 37, 21), (60, 40, 10, 10), (0, 0, 70, 50), (33, 0, 1, 50)):
        sc = oracle.scene(hostmirror.cbox_scene(70, 50, crop=crop))
        for prm in (abi.render_params(spp=2, seed=2), abi.render_params(spp=2, seed=2, block_size=16, block_first=1, block_stride=3),
                    abi.render_params(spp=1, rng_mode=abi.MSK_RNG_PCG_BLOCK)):
            a, sa = sc.render(prm, threads=4)
            b, sb = full.render(prm, threads=4)
            x, y, w, h = crop
            assert a.shape == (h, w, 5) and np.array_equal(a.view(np.uint32), b[y:y + h, x:x + w].view(np.uint32))
            assert sa.samples <= sb.samples
        sc.close()
    full.close()


def test_oracle_renders_degenerate_scenes(oracle, abi, hostmirror):
    for name, (meshes, env) in degenerate_scenes(hostmirror).items():
        sc = oracle.scene(hostmirror.flatten(meshes, 24, 16, env=env))
        film, st = sc.render(abi.render_params(4, seed=1), threads=2)
        sc.close()
        assert np.isfinite(film).all() and film[..., 4].min() > 0 and st.samples == 24 * 16 * 4, name
        lit = film[..., :3].sum() > 0
        assert lit == (name in ("environment only", "mesh without faces", "one emitting triangle")), name
