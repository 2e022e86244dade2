"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle.  Need an MI355X.

Bar: bit-exact for indices AND for every fp32 value (hit records, per-sample XYZ, film sums);
the fp32 tolerance the north star allows (1e-4 per pixel) is only used for cross-shard sums.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cbox(hostmirror, golden_lookup, w, h, extra=()):
    # the product's own srgb_model_fetch; golden_lookup is that function plus a bit-equality check against the
    # reference-fetched coefficients for the colours that were recorded (tests/conftest.py)
    return hostmirror.cbox_scene(w, h, coeff_lookup=golden_lookup, extra_meshes=extra)


@pytest.fixture(scope="module")
def scene256(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    flat = cbox(hostmirror, golden_lookup, 256, 256)
    g = abi.Scene(gpu_ctx, flat)
    o = oracle.scene(flat)
    yield g, o, flat
    g.close()
    o.close()


def random_rays(flat, n, seed):
    rng = np.random.RandomState(seed)
    v = flat.vertices[:, :3]
    lo, hi = v.min(0), v.max(0)
    o = rng.uniform(lo - 50, hi + 50, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1e-3, d, np.inf
    # a third of the rays get a finite far bound, some start on a surface (secondary-ray like)
    rays[::3, 7] = rng.uniform(10, 900, len(rays[::3])).astype(np.float32)
    return rays


def camera_rays(oscene, w, h, n, seed):
    rng = np.random.RandomState(seed)
    return np.stack([oscene.camera_ray(0.5, *p)[0] for p in rng.uniform(0, [w, h], (n, 2)).astype(np.float32)])


def test_library_is_the_hip_build(gpu_ctx, abi):
    assert abi.LIB_PATH.endswith("misaki-render_amd/lib/libmsk_gpu.so")
    assert "gfx950" in gpu_ctx.describe()


def test_trace_closest_and_any_bit_exact(scene256):
    g, o, flat = scene256
    rays = np.concatenate([random_rays(flat, 400_000, 1), camera_rays(o, 256, 256, 2000, 2)])
    # secondary rays: start exactly on hit points of the first batch
    o.set_bvh(1)
    first = o.trace_closest(rays)
    ok = np.isfinite(first[:, 0])
    p = rays[ok, :3] + rays[ok, 4:7] * first[ok, 0:1]
    sec = random_rays(flat, int(ok.sum()), 3)
    sec[:, :3] = p
    sec[:, 3] = (1 + np.abs(p).max(1)) * np.float32(8.940697e-05)
    rays = np.concatenate([rays, sec])
    want = o.trace_closest(rays)
    got = g.trace_closest(rays)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    # and the oracle's BVH answers == its brute force answers (order independence of the hit rule)
    o.set_bvh(0)
    sub = rays[::7]
    assert np.array_equal(o.trace_closest(sub).view(np.uint32), want[::7].view(np.uint32))
    o.set_bvh(1)
    hits = np.isfinite(want[:, 0]).mean()
    assert 0.3 < hits < 1.0


def test_trace_degenerate_rays(scene256):
    g, o, _ = scene256
    rays = np.array([[278, 273, -800, 0, 0, 0, 0, np.inf],          # zero direction (failed BSDF sample)
                     [278, 273, 100, 1e-3, 0, 1, 0, np.inf],         # axis aligned, hits the light
                     [278, 273, 100, 1e-3, 0, -1, 0, np.inf],
                     [278, 273, 100, 1e-3, 1, 0, 0, 10.0],           # far bound before the wall
                     [0, 0, 0, 1e-3, 0, 0, 1, np.inf],               # along a shared edge / corner
                     [130, 165, 65, 1e-3, 0, 1, 0, np.inf]], np.float32)
    assert np.array_equal(g.trace_closest(rays).view(np.uint32), o.trace_closest(rays).view(np.uint32))
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    assert g.trace_closest(np.zeros((0, 8), np.float32)).shape == (0, 4)


def test_per_sample_radiance_bit_exact(scene256, abi):
    g, o, _ = scene256
    prm = abi.render_params(spp=64, seed=3)
    rng = np.random.RandomState(5)
    pixels = np.concatenate([rng.randint(0, 256, (48, 2)), [[128, 30], [0, 0], [255, 255], [128, 128]]]).astype(np.int32)
    gx, gp = g.sample_pixels(prm, pixels)
    ox, op = o.sample_pixels(prm, pixels)
    assert np.array_equal(gp, op)
    assert np.array_equal(gx.view(np.uint32), ox.view(np.uint32))
    assert np.isfinite(gx).all() and gx.max() > 1.0       # the light is in view of some samples


@pytest.mark.parametrize("kw", [dict(), dict(max_depth=1), dict(max_depth=2), dict(max_depth=0), dict(rr_depth=1),
                                dict(rr_depth=50, max_depth=12), dict(hide_emitters=1), dict(seed=123456789012345)])
def test_integrator_properties_bit_exact(scene256, abi, kw):
    g, o, _ = scene256
    prm = abi.render_params(spp=8, **kw)
    pixels = np.array([[128, 30], [100, 200], [30, 128], [220, 128], [128, 128], [90, 140]], np.int32)
    gx, _ = g.sample_pixels(prm, pixels)
    ox, _ = o.sample_pixels(prm, pixels)
    assert np.array_equal(gx.view(np.uint32), ox.view(np.uint32))


def test_c1_film_bit_exact(scene256, abi, hostmirror):
    """BASELINE config 1: cbox 256x256 @ 16 spp, GPU film == oracle film, bit for bit."""
    g, o, _ = scene256
    prm = abi.render_params(spp=16, seed=0)
    film, st = g.render(prm)
    ref, rst = o.render(prm, threads=8)
    # segment statistics: the GPU drops a zero-throughput path one ray earlier than the scalar loop (same result)
    assert st.samples == 256 * 256 * 16 and abs(int(st.segments) - int(rst.segments)) <= 1e-5 * rst.segments
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    rgb, rgb_ref = hostmirror.develop(film), hostmirror.develop(ref)
    assert np.sqrt(((rgb[..., :3] - rgb_ref[..., :3]) ** 2).sum(-1)).max() == 0.0
    assert np.all(film[..., 4] > 0) and np.array_equal(film[..., 3], film[..., 4])     # A == W


def test_ragged_film_and_single_sample(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    flat = cbox(hostmirror, golden_lookup, 100, 40)        # blocks of 32: ragged right and bottom edges
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    for spp in (1, 5):
        prm = abi.render_params(spp=spp, seed=9)
        film, _ = g.render(prm)
        ref, _ = o.render(prm, threads=4)
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    for bs in (16, 48, 96, 200):         # integrator.cpp:20 takes any block size: one block covers the whole film from 100 on
        prm = abi.render_params(spp=3, block_size=bs)
        assert np.array_equal(g.render(prm)[0].view(np.uint32), o.render(prm, threads=4)[0].view(np.uint32)), bs
    g.close()
    o.close()


def test_crop_window_bit_exact(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """film.cpp:12-21 / hdrfilm.cpp:37-38: the film's crop window.  The sensor and the block schedule stay those of the full
    100x40 film; the film that comes back is the 37x21 window at (11, 5) — bit for bit the oracle's, and bit for bit that
    window of the uncropped film (Film::put clips every block to the storage, imageblock.cpp:133-173)."""
    full = cbox(hostmirror, golden_lookup, 100, 40)
    crop = hostmirror.cbox_scene(100, 40, coeff_lookup=golden_lookup, crop=(11, 5, 37, 21))
    g, gf, o = abi.Scene(gpu_ctx, crop), abi.Scene(gpu_ctx, full), oracle.scene(crop)
    for prm in (abi.render_params(spp=6, seed=3), abi.render_params(spp=3, seed=3, block_size=16), abi.render_params(spp=4, block_first=1, block_stride=2)):
        film, st = g.render(prm)
        ref, rst = o.render(prm, threads=4)
        assert film.shape == (21, 37, 5) and st.samples == rst.samples and st.segments == rst.segments
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
        whole, wst = gf.render(prm)
        assert np.array_equal(film.view(np.uint32), whole[5:26, 11:48].view(np.uint32))
        assert st.samples < wst.samples                           # the blocks that cannot reach the window are not rendered
    # a window in the last, ragged block row and column; the whole film given explicitly
    for c in ((90, 30, 10, 10), (0, 0, 100, 40), (64, 0, 1, 1)):
        fc = hostmirror.cbox_scene(100, 40, coeff_lookup=golden_lookup, crop=c)
        gc = abi.Scene(gpu_ctx, fc)
        prm = abi.render_params(spp=2, seed=5)
        film, _ = gc.render(prm)
        whole, _ = gf.render(prm)
        assert np.array_equal(film.view(np.uint32), whole[c[1]:c[1] + c[3], c[0]:c[0] + c[2]].view(np.uint32)), c
        gc.close()
    for bad in ((-1, 0, 10, 10), (0, 0, 101, 40), (95, 35, 10, 10), (5, 5, 0, 7)):
        with pytest.raises(abi.MskError) as e:
            abi.Scene(gpu_ctx, hostmirror.cbox_scene(100, 40, coeff_lookup=golden_lookup, crop=bad))
        assert "crop" in str(e.value)
    for x in (g, gf, o):
        x.close()


def test_shards_sum_to_the_whole(scene256, abi, hostmirror):
    g, o, _ = scene256
    full, _ = g.render(abi.render_params(spp=8, seed=4))
    # pixel-tile shard (the multi-GPU decomposition): every rank's film is bit-exact vs the oracle shard
    parts = []
    for r in range(3):
        prm = abi.render_params(spp=8, seed=4, block_first=r, block_stride=3)
        f, st = g.render(prm)
        assert np.array_equal(f.view(np.uint32), o.render(prm, threads=8)[0].view(np.uint32))
        parts.append(f)
    s = parts[0] + parts[1] + parts[2]
    interior = np.ones((256, 256), bool)
    for k in range(0, 256, 32):                       # pixels within the 2-px filter border of a tile edge
        interior[max(0, k - 2):k + 2, :] = False
        interior[:, max(0, k - 2):k + 2] = False
    assert np.array_equal(s[interior], full[interior])           # one rank contributes: exact
    assert np.allclose(s, full, rtol=3e-7, atol=1e-6)             # <= 4 ranks contribute: fp32 re-association
    d = hostmirror.develop(s)[..., :3] - hostmirror.develop(full)[..., :3]
    assert np.sqrt((d ** 2).sum(-1)).max() < 1e-4
    # sample-index shard
    sp = [g.render(abi.render_params(spp=8, seed=4, sample_first=r, sample_stride=2))[0] for r in range(2)]
    assert np.allclose(sp[0] + sp[1], full, rtol=2e-5, atol=1e-5)
    assert np.array_equal(sp[1].view(np.uint32),
                          o.render(abi.render_params(spp=8, seed=4, sample_first=1, sample_stride=2), threads=8)[0].view(np.uint32))


def test_large_mesh_uses_hbm_bvh(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """A 20k-triangle blob inside the box: BVH no longer fits LDS; also exercises deep traversal."""
    blob = hostmirror.blob_mesh("blob", (370, 420, 250), 70, 100, 100, hostmirror.WHITE, seed=2)
    flat = cbox(hostmirror, golden_lookup, 96, 96, extra=[blob])
    assert flat.desc.n_faces > 19000
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    rays = np.concatenate([random_rays(flat, 200_000, 11), camera_rays(o, 96, 96, 1000, 12)])
    want = o.trace_closest(rays)
    assert np.array_equal(g.trace_closest(rays).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    blob_hits = (want[:, 3].view(np.uint32) >= 32) & np.isfinite(want[:, 0])
    assert blob_hits.sum() > 1000
    prm = abi.render_params(spp=4, seed=2)
    film, _ = g.render(prm)
    ref, _ = o.render(prm, threads=8)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    g.close()
    o.close()


@pytest.mark.parametrize("env", [dict(MSK_TRACE_REFILL="0"), dict(MSK_WIDE_BVH="0"), dict(MSK_WIDE_BVH="0", MSK_TRACE_REFILL="0"),
                                 dict(MSK_TRACE_REFILL="48", MSK_TRACE_QUANTUM="1"), dict(MSK_LDS_SCENE_KB="0", MSK_WIDE_BVH="0"),
                                 dict(MSK_STACK_CAP="4"), dict(MSK_STACK_CAP="4", MSK_TRACE_REFILL="0", MSK_WIDE_BVH="0"),
                                 dict(MSK_COLLAPSE_OPTIMAL="0"), dict(MSK_COLLAPSE_OPTIMAL="0", MSK_QUANT_BVH="0", MSK_TRACE_REFILL="0"),
                                 dict(MSK_WIDE_LDS="1", MSK_TRACE_REFILL="0"),
                                 dict(MSK_QUANT_BVH="0"), dict(MSK_QUANT_BVH="0", MSK_TRACE_REFILL="0"), dict(MSK_QUANT_BVH="0", MSK_STACK_CAP="4"),
                                 dict(MSK_BVH_BUILD="gpu", MSK_QUANT_BVH="0"),
                                 dict(MSK_QUANT_BVH="1"), dict(MSK_QUANT_BVH="1", MSK_TRACE_REFILL="0"), dict(MSK_QUANT_BVH="1", MSK_BVH_BUILD="gpu", MSK_STACK_CAP="4"),
                                 dict(MSK_QUANT_BVH="1", MSK_TRACE_QUANTUM="4", MSK_BVH_SWEEP="0"), dict(MSK_BVH_SWEEP="0"), dict(MSK_BVH_SWEEP="64", MSK_WIDE_BVH="0"),
                                 dict(MSK_WIDE_BVH="8"), dict(MSK_WIDE_BVH="8", MSK_TRACE_REFILL="0"),
                                 dict(MSK_WIDE_BVH="8", MSK_STACK_CAP="4", MSK_TRACE_QUANTUM="2"),
                                 dict(MSK_BVH_BUILD="gpu"), dict(MSK_BVH_BUILD="gpu", MSK_WIDE_BVH="0", MSK_STACK_CAP="4"),
                                 dict(MSK_BVH_BUILD="gpu", MSK_WIDE_BVH="8"), dict(MSK_BVH_BUILD="gpu", MSK_BVH_LEAF="1", MSK_TRACE_REFILL="0")])
def test_every_traversal_kernel_gives_the_same_film(gpu_ctx, abi, hostmirror, oracle, golden_lookup, monkeypatch, env):
    """k_trace<0|1|2|4|5|6> (chunk loop) and k_trace_r<0|1|2|4|5|6> (lane replacement), binary, 4-wide (80-byte nodes with half-float
    boxes read by v_fma_mix_f32: the default for trees in HBM since round 5; MSK_QUANT_BVH=1: rounds 3-4's 64-byte byte-quantised
    nodes; MSK_QUANT_BVH=0: the full-precision 128-byte ones) and 8-wide quantised trees, built by the host's SAH builder (a full
    sweep on the first MSK_BVH_SWEEP levels, 16 bins below) or on the device (MSK_BVH_BUILD=gpu, msk_lbvh.hip): hit selection is by (t, prim), so
    every one of them must reproduce the oracle's film bit for bit."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    blob = hostmirror.blob_mesh("blob", (370, 420, 250), 70, 40, 40, hostmirror.WHITE, seed=2)
    big = cbox(hostmirror, golden_lookup, 64, 64, extra=[blob])
    small = cbox(hostmirror, golden_lookup, 64, 64)
    prm = abi.render_params(spp=4, seed=6)
    for flat, extra_env in ((big, {}), (small, dict(MSK_TRACE_REFILL=env.get("MSK_TRACE_REFILL", "16")))):
        for k, v in extra_env.items():
            monkeypatch.setenv(k, v)                  # the LDS-resident scene only uses k_trace_r when asked to
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        film, _ = g.render(prm)
        ref, _ = o.render(prm, threads=8)
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), (env, flat.desc.n_faces)
        g.close()
        o.close()


def test_full_hd_ragged_tiles_bit_exact(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """BASELINE config 4's film (1920x1080: 60 x 34 tiles, the last row 24 px high, imageblock.cpp:206-208) at 2 spp."""
    flat = cbox(hostmirror, golden_lookup, 1920, 1080)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    prm = abi.render_params(spp=2, seed=1)
    film, st = g.render(prm)
    ref, rst = o.render(prm, threads=8)
    assert st.samples == rst.samples == 1920 * 1080 * 2
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    g.close()
    o.close()


def test_headline_size_properties(gpu_ctx, abi, hostmirror, golden_lookup):
    """BASELINE config 2 at full size (512^2 x 512 spp, too long for the scalar oracle): size-independent properties.
    (i) the weight channels depend on the film positions only, so they equal those of a depth-0 render bit for bit;
    (ii) the film is linear in the sample set: two sample-index shards sum to the whole; (iii) a second run is identical."""
    flat = cbox(hostmirror, golden_lookup, 512, 512)
    g = abi.Scene(gpu_ctx, flat)
    full, st = g.render(abi.render_params(spp=512, seed=3))
    assert st.samples == 512 * 512 * 512 and np.isfinite(full).all() and full[..., :3].min() >= 0
    flat0, _ = g.render(abi.render_params(spp=512, seed=3, max_depth=0))
    assert np.array_equal(full[..., 3:].view(np.uint32), flat0[..., 3:].view(np.uint32)) and not flat0[..., :3].any()
    assert np.array_equal(full[..., 3], full[..., 4])
    halves = [g.render(abi.render_params(spp=512, seed=3, sample_first=r, sample_stride=2))[0] for r in range(2)]
    assert np.allclose(halves[0] + halves[1], full, rtol=1e-4, atol=1e-4)
    again, _ = g.render(abi.render_params(spp=512, seed=3))
    assert np.array_equal(again.view(np.uint32), full.view(np.uint32))
    g.close()


def test_config2_full_size_film_bit_exact(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """BASELINE config 2 at FULL size — cbox 512x512 @ 512 spp, counter RNG, seed 0 (134 M samples; the oracle takes 20-30 s on
    the box's host threads): the GPU film equals the oracle's film bit for bit, hence per-pixel L2 of the developed image = 0
    (the north star allows 1e-4)."""
    import os
    flat = cbox(hostmirror, golden_lookup, 512, 512)
    prm = abi.render_params(spp=512, seed=0)
    g = abi.Scene(gpu_ctx, flat)
    film, st = g.render(prm)
    g.close()
    o = oracle.scene(flat)
    ref, rst = o.render(prm, threads=len(os.sched_getaffinity(0)))
    o.close()
    assert st.samples == rst.samples == 512 * 512 * 512
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    a, b = hostmirror.develop(film)[..., :3].astype(np.float64), hostmirror.develop(ref)[..., :3].astype(np.float64)
    assert np.sqrt(((a - b) ** 2).sum(-1)).max() == 0.0


def test_two_emitters(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """Multi-emitter selection path of Scene::sample_emitter_direct (scene.cpp:78-88)."""
    second = hostmirror.MeshSpec("lamp2", [((100, 300, 558), (200, 300, 558), (200, 400, 558), (100, 400, 558))],
                                 hostmirror.WHITE, radiance=(40, 40, 40))
    flat = cbox(hostmirror, golden_lookup, 64, 64, extra=[second])
    assert flat.desc.n_emitters == 2
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    prm = abi.render_params(spp=16, seed=8)
    film, _ = g.render(prm)
    ref, _ = o.render(prm, threads=8)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    g.close()
    o.close()


def test_multi_pass_record_budget(scene256, abi, monkeypatch):
    """Forcing a tiny record budget splits the render into several block passes; same film."""
    g, o, _ = scene256
    prm = abi.render_params(spp=4, seed=6)
    one, st1 = g.render(prm)
    monkeypatch.setenv("MSK_RECORD_BUDGET_MB", "1")
    many, st2 = g.render(prm)
    assert st2.passes > st1.passes == 1
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))


def test_pool_shape_does_not_change_results(scene256, abi, monkeypatch):
    g, o, _ = scene256
    prm = abi.render_params(spp=4, seed=6)
    a, _ = g.render(prm)
    monkeypatch.setenv("MSK_REGIONS", "64")
    monkeypatch.setenv("MSK_REGION_SIZE", "128")
    b, st = g.render(prm)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and st.iterations > 20


def test_error_behaviour(scene256, gpu_ctx, abi, hostmirror, golden_lookup):
    g, o, flat = scene256
    with pytest.raises(abi.MskError) as e:
        g.render(abi.render_params(spp=4, rng_mode=5))            # neither MSK_RNG_COUNTER nor MSK_RNG_PCG_BLOCK (tests/test_pcg_block.py)
    assert e.value.code == abi.MSK_ERR_INVALID_ARG
    with pytest.raises(abi.MskError) as e:
        g.render(abi.render_params(spp=4, rr_depth=0))            # integrator.cpp:131-132
    assert "rr_depth" in str(e.value)
    with pytest.raises(abi.MskError):
        g.render(abi.render_params(spp=4, max_depth=-2))          # integrator.cpp:135-136
    with pytest.raises(abi.MskError):
        g.render(abi.render_params(spp=0))
    # the path state holds bounces up to 4094: a bound or a roulette start beyond that is refused, not silently cut
    with pytest.raises(abi.MskError) as e:
        g.render(abi.render_params(spp=1, max_depth=5000))
    assert e.value.code == abi.MSK_ERR_UNSUPPORTED and "max_depth" in str(e.value)
    with pytest.raises(abi.MskError) as e:
        g.render(abi.render_params(spp=1, rr_depth=5000))
    assert e.value.code == abi.MSK_ERR_UNSUPPORTED and "rr_depth" in str(e.value)
    g.render(abi.render_params(spp=1, rr_depth=5000, max_depth=12))       # bounded: fine
    bad = cbox(hostmirror, golden_lookup, 32, 32)
    bad.desc.meshes[2].bsdf_id = 99
    with pytest.raises(abi.MskError) as e:
        abi.Scene(gpu_ctx, bad)
    assert e.value.code == abi.MSK_ERR_INVALID_ARG
    bad = cbox(hostmirror, golden_lookup, 32, 32)
    bad.faces[5, 1] = 1000
    with pytest.raises(abi.MskError):
        abi.Scene(gpu_ctx, bad)
    with pytest.raises(abi.MskError):
        g.sample_pixels(abi.render_params(spp=1), np.array([[300, 2]], np.int32))


def test_two_members_behind_one_context(abi, hostmirror, oracle, golden_lookup):
    """msk_gpu_init(ids, n = 2) — rehearsed on the one GPU of the box with ids = {0, 0}: two member contexts, each renders the
    sample indices s = k (mod 2), the films are summed on the first device by k_film_sum.  The union of the shards is the
    single-context sample set, so statistics are equal and the film differs from the single-context (= oracle) film only by
    the re-association of two partial sums per pixel."""
    flat = cbox(hostmirror, golden_lookup, 96, 64)
    prm = abi.render_params(spp=9, seed=5)                     # odd: the members own 5 and 4 samples per pixel
    with abi.Context(0) as one:
        s1 = abi.Scene(one, flat)
        ref, st1 = s1.render(prm)
        s1.close()
    with abi.Context((0, 0)) as grp:
        assert "2 devices [0,0]" in grp.describe()
        s2 = abi.Scene(grp, flat)
        film, st2 = s2.render(prm)
        film_b, _ = s2.render(prm)
        # a caller-side shard of the samples composes with the members' split
        half_a, _ = s2.render(abi.render_params(spp=9, seed=5, sample_first=0, sample_stride=2))
        half_b, _ = s2.render(abi.render_params(spp=9, seed=5, sample_first=1, sample_stride=2))
        one_spp, st_one = s2.render(abi.render_params(spp=1, seed=5))          # the second member owns nothing
        aov, _ = s2.render_aov(prm, [abi.MSK_AOV_DEPTH, abi.MSK_AOV_PATH_RGBA])
        px = np.array([[3, 4], [50, 60]], np.int32)
        xyz, _ = s2.sample_pixels(abi.render_params(spp=4, seed=5), px)
        with pytest.raises(abi.MskError) as e:
            s2.render(abi.render_params(spp=0))
        assert "member 0 of 2" in str(e.value)
        s2.close()
    assert (st2.samples, st2.segments, st2.shadow_rays) == (st1.samples, st1.segments, st1.shadow_rays)
    assert np.array_equal(film, film_b)
    assert np.allclose(film, ref, rtol=2e-6, atol=1e-6) and np.array_equal(film[..., 4].sum(dtype=np.float64) > 0, True)
    a, b = hostmirror.develop(film)[..., :3], hostmirror.develop(ref)[..., :3]
    assert np.abs(a - b).max() < 1e-4
    assert np.allclose(half_a + half_b, ref, rtol=2e-6, atol=1e-6)
    assert st_one.samples == 96 * 64 and np.isfinite(one_spp).all()
    assert aov.shape[-1] == 5 + 1 + 4 and np.allclose(aov[..., :3], ref[..., :3], rtol=2e-6, atol=1e-6)
    o = oracle.scene(flat)
    xyz_ref, _ = o.sample_pixels(abi.render_params(spp=4, seed=5), px)
    o.close()
    assert np.array_equal(xyz.view(np.uint32), xyz_ref.view(np.uint32))
    with pytest.raises(abi.MskError):
        abi.Context([0] * 9)


def test_group_film_sum_staged_branch_equals_peer_branch(abi, hostmirror, golden_lookup, monkeypatch, capfd):
    """The two ways a member's film reaches the first device (msk_multi.h): read in place by k_film_sum (peer access — what
    ids = {0, 0, 0} always gets) or copied into a staging buffer by hipMemcpyPeerAsync first (a member whose memory cannot be
    mapped).  MSK_GROUP_FORCE_STAGED=1 takes the second branch on a one-GPU box: same members, same order of the sum —
    the films must be equal bit for bit; MSK_GROUP_LOG=1 says per member which branch it got (HDRFilm::put's mutexed
    accumulation, films/hdrfilm.cpp:43-46, is what both replace)."""
    flat = cbox(hostmirror, golden_lookup, 80, 48)
    prm = abi.render_params(spp=7, seed=3)
    films = {}
    for forced in ("0", "1"):
        monkeypatch.setenv("MSK_GROUP_FORCE_STAGED", forced)
        monkeypatch.setenv("MSK_GROUP_LOG", "1")
        with abi.Context((0, 0, 0)) as grp:
            sc = abi.Scene(grp, flat)
            films[forced], st = sc.render(prm)
            again, _ = sc.render(prm)
            sc.close()
        assert st.samples == 80 * 48 * 7 and np.array_equal(again, films[forced])
        err = capfd.readouterr().err
        lines = [l for l in err.splitlines() if l.startswith("[msk_gpu] group member")]
        assert len(lines) == 2, err
        assert all(("staged hipMemcpyPeerAsync" in l and "MSK_GROUP_FORCE_STAGED" in l) if forced == "1" else ("in place over peer access" in l and "same device" in l)
                   for l in lines), lines
    assert np.array_equal(films["0"].view(np.uint32), films["1"].view(np.uint32))
    assert np.isfinite(films["0"]).all() and films["0"][..., :3].max() > 0


def test_scene_lifecycle_releases_device_memory(gpu_ctx, abi, hostmirror, golden_lookup):
    """create / render / destroy in a loop: every DevBuf and the cached workspace go back to the allocator, a second
    context on the same device works side by side, and results do not depend on what ran before."""
    import torch
    flat = cbox(hostmirror, golden_lookup, 128, 96)
    prm = abi.render_params(spp=4, seed=9)
    first = None
    free0 = None
    for i in range(6):
        g = abi.Scene(gpu_ctx, flat)
        film, _ = g.render(prm)
        if i % 3 == 0:
            g.render_aov(prm, [abi.MSK_AOV_DEPTH, abi.MSK_AOV_PATH_RGBA])
        g.close()
        if first is None:
            first = film
        assert np.array_equal(film.view(np.uint32), first.view(np.uint32))
        free, total = torch.cuda.mem_get_info(0)
        if i == 1:
            free0 = free
        if i > 1:
            assert free >= free0 - (64 << 20), (i, free0, free)          # no growth from iteration to iteration
    other = abi.Context(0)
    g2 = abi.Scene(other, flat)
    assert np.array_equal(g2.render(prm)[0].view(np.uint32), first.view(np.uint32))
    g2.close()
    other.close()


@pytest.mark.parametrize("stddev,block", [(0.3, 8), (0.625, 16), (1.0, 32), (1.7, 32), (0.125, 4)])
def test_other_filter_widths_bit_exact(gpu_ctx, oracle, abi, hostmirror, golden_lookup, stddev, block):
    """GaussianFilter's stddev property (gaussian.cpp:12-14): radius = 4 stddev, border = ceil(radius - .5) (rfilter.cpp:22)
    — the film replay's gather window and the block borders follow the radius."""
    flat = hostmirror.flatten(hostmirror.cbox_meshes(), 83, 61, coeff_lookup=golden_lookup, filter_stddev=stddev)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    prm = abi.render_params(spp=5, seed=4, block_size=block)
    film, st = g.render(prm)
    ref, rst = o.render(prm, threads=8)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), float(np.abs(film - ref).max())
    if 2 * int(np.ceil(4 * stddev - .5)) > 4:
        with pytest.raises(abi.MskError):                      # a block must hold its neighbours' borders
            g.render(abi.render_params(spp=1, block_size=4))
    g.close(); o.close()


def test_sample_ranges_bit_exact_and_sum_to_the_whole(scene256, abi, hostmirror):
    """msk_render_params: s = sample_first + k * sample_stride below spp — contiguous ranges (stride 1) and offsets beyond
    the stride, the selectors the speed-proportional multi-GPU split uses."""
    g, o, flat = scene256
    whole, _ = g.render(abi.render_params(spp=9, seed=2))
    acc = np.zeros_like(whole, dtype=np.float64)
    n = 0
    for first, stride, spp in ((0, 1, 4), (4, 1, 7), (7, 1, 9)):
        prm = abi.render_params(spp=spp, seed=2, sample_first=first, sample_stride=stride)
        film, st = g.render(prm)
        ref, rst = o.render(prm, threads=8)
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), (first, stride, spp)
        assert st.samples == rst.samples == (spp - first) * 256 * 256
        acc += film
        n += st.samples
    assert n == 9 * 256 * 256 and np.allclose(acc, whole, rtol=2e-6, atol=1e-6)
    prm = abi.render_params(spp=12, seed=2, sample_first=5, sample_stride=3)           # s = 5, 8, 11
    film, st = g.render(prm)
    ref, rst = o.render(prm, threads=8)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)) and st.samples == rst.samples == 3 * 256 * 256
    film, st = g.render(abi.render_params(spp=4, seed=2, sample_first=4))              # empty range: zeros, no error
    assert st.samples == 0 and not film.any()


def test_degenerate_scenes_bit_exact(gpu_ctx, oracle, abi, hostmirror):
    from test_oracle_kat import degenerate_scenes
    for name, (meshes, env) in degenerate_scenes(hostmirror).items():
        flat = hostmirror.flatten(meshes, 24, 16, env=env)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        for kw in (dict(), dict(hide_emitters=1), dict(max_depth=1)):
            prm = abi.render_params(4, seed=1, **kw)
            film, st = g.render(prm)
            ref, rst = o.render(prm, threads=2)
            assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), (name, kw)
            assert st.samples == rst.samples
        g.close(); o.close()


def test_copy_back_into_pinned_and_pageable_targets_is_the_same_film(gpu_ctx, abi, hostmirror, golden_lookup, monkeypatch):
    """msk_gpu_render's film copy-back: a caller's array in pinned host memory takes the DMA directly, a pageable one goes through
    the library's staging buffer (msk_gpu.hip) — the same bytes arrive either way, also with the direct path switched off."""
    import torch
    flat = cbox(hostmirror, golden_lookup, 120, 72)
    sc = abi.Scene(gpu_ctx, flat)
    prm = abi.render_params(spp=5, seed=12)
    pageable, _ = sc.render(prm)
    pinned = torch.full((72, 120, 5), -1.0, dtype=torch.float32).pin_memory().numpy()
    out, st = sc.render(prm, out=pinned)
    assert out is pinned and st.samples == 120 * 72 * 5 and np.array_equal(pinned.view(np.uint32), pageable.view(np.uint32))
    monkeypatch.setenv("MSK_COPYBACK_STAGED", "1")
    pinned[:] = -1.0
    sc.render(prm, out=pinned)
    assert np.array_equal(pinned.view(np.uint32), pageable.view(np.uint32))
    # a view into the middle of a pinned allocation is pinned memory too
    big = torch.zeros((3, 72, 120, 5), dtype=torch.float32).pin_memory().numpy()
    monkeypatch.delenv("MSK_COPYBACK_STAGED")
    sc.render(prm, out=big[1])
    assert np.array_equal(big[1].view(np.uint32), pageable.view(np.uint32)) and not big[0].any() and not big[2].any()
    sc.close()
