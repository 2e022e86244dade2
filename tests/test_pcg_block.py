"""MSK_RNG_PCG_BLOCK on the device (csrc/msk_serial.h): the reference's sampler semantics as written — ONE PCG32 stream per image
block (samplers/independent.cpp:9-35, integrator.cpp:56-58), a block's samples drawing from it in the scalar loops' order — so one
lane renders one block, path after path.  The oracle implements the same mode (oracle.cpp: Sampler); the films must agree bit
for bit, which also pins how many draws every path consumed (the next sample starts where the last path stopped)."""
import numpy as np
import pytest

from test_environment import open_box_scene
from test_rough_conductor import conductor_scene
from test_rough_dielectric import glass_scene

pytestmark = pytest.mark.gpu


def _same(abi, g, o, **kw):
    prm = abi.render_params(rng_mode=abi.MSK_RNG_PCG_BLOCK, **kw)
    film, st = g.render(prm)
    ref, rst = o.render(prm, threads=8)
    assert st.samples == rst.samples and st.segments == rst.segments
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)), kw
    return film


def test_baseline_config_1_in_the_references_sampler_mode(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    """BASELINE config 1: assets/cbox at 256x256, 16 spp, path integrator — with the per-block PCG32 stream the reference's
    `independent` sampler prescribes: the GPU film is the oracle's, bit for bit."""
    flat = hostmirror.cbox_scene(256, 256, coeff_lookup=golden_lookup)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    film = _same(abi, g, o, spp=16)
    assert np.isfinite(film).all() and np.array_equal(film[..., 3], film[..., 4]) and film[..., 4].min() > 0
    # and it is a different estimate of the same image than the counter-RNG film (same spp): close, not equal
    cf, _ = g.render(abi.render_params(spp=16))
    a, b = hostmirror.develop(film)[..., :3], hostmirror.develop(cf)[..., :3]
    assert not np.array_equal(film, cf) and abs(a.mean() - b.mean()) < 0.02 * b.mean()
    g.close()
    o.close()


def test_integrator_properties_shards_crop_and_ragged_blocks(gpu_ctx, abi, hostmirror, oracle, golden_lookup):
    flat = hostmirror.cbox_scene(100, 40, coeff_lookup=golden_lookup)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    for kw in (dict(spp=5, seed=3), dict(spp=3, rr_depth=1), dict(spp=3, rr_depth=2, max_depth=3), dict(spp=2, max_depth=1), dict(spp=2, hide_emitters=1),
               dict(spp=3, block_size=16), dict(spp=2, block_size=200), dict(spp=3, block_first=1, block_stride=3),
               dict(spp=4, block_size=16, block_first=2, block_stride=5)):
        _same(abi, g, o, **kw)
    # a shard of the SAMPLE indices is refused: a block's samples share one sequential stream (every shard would draw the same numbers)
    for kw in (dict(spp=6, sample_first=1, sample_stride=2), dict(spp=5, sample_first=2, sample_stride=1)):
        with pytest.raises(abi.MskError) as e:
            g.render(abi.render_params(rng_mode=abi.MSK_RNG_PCG_BLOCK, **kw))
        assert e.value.code == abi.MSK_ERR_UNSUPPORTED and "block_first" in str(e.value)
    g.close()
    o.close()
    crop = hostmirror.cbox_scene(100, 40, coeff_lookup=golden_lookup, crop=(11, 5, 37, 21))
    g, o = abi.Scene(gpu_ctx, crop), oracle.scene(crop)
    assert _same(abi, g, o, spp=3).shape == (21, 37, 5)
    g.close()
    o.close()


@pytest.mark.parametrize("scene", ["conductor", "glass", "open_box"])
def test_every_bsdf_and_emitter_type(gpu_ctx, abi, hostmirror, oracle, golden_lookup, scene):
    """Rough conductors (one- and two-sided), a rough dielectric mesh (a tree in HBM for the wavefront path; the serial path
    walks the binary tree), area + environment emitters, paths that leave the scene: the draws a failed or zero-weight BSDF
    sample still consumes are part of what must agree."""
    make = {"conductor": conductor_scene, "glass": glass_scene, "open_box": open_box_scene}[scene]
    flat = make(hostmirror, golden_lookup, 48, 40)
    g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
    _same(abi, g, o, spp=4, seed=2)
    _same(abi, g, o, spp=2, rr_depth=2, max_depth=6, hide_emitters=1)
    g.close()
    o.close()


def test_on_a_device_built_tree(gpu_ctx, abi, hostmirror, oracle, golden_lookup, monkeypatch):
    """MSK_BVH_BUILD=gpu: the linear BVH of msk_lbvh.hip is deeper than the host builder's tree and k_path_serial sizes its
    traversal stack by the tree's depth (round 4: the depth of a device-built tree reached it as 0 — a GPU memory fault)."""
    monkeypatch.setenv("MSK_BVH_BUILD", "gpu")
    for flat in (hostmirror.cbox_scene(64, 48, coeff_lookup=golden_lookup), glass_scene(hostmirror, golden_lookup, 48, 48)):
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        _same(abi, g, o, spp=3)
        g.close()
        o.close()


def test_what_the_mode_does_not_cover(gpu_ctx, abi, hostmirror, golden_lookup):
    g = abi.Scene(gpu_ctx, hostmirror.cbox_scene(32, 32, coeff_lookup=golden_lookup))
    with pytest.raises(abi.MskError) as e:
        g.sample_pixels(abi.render_params(spp=2, rng_mode=abi.MSK_RNG_PCG_BLOCK), [[3, 4]])
    assert e.value.code == abi.MSK_ERR_UNSUPPORTED
    with pytest.raises(abi.MskError) as e:
        g.render_aov(abi.render_params(spp=2, rng_mode=abi.MSK_RNG_PCG_BLOCK), [abi.MSK_AOV_DEPTH])
    assert e.value.code == abi.MSK_ERR_UNSUPPORTED
    with pytest.raises(abi.MskError) as e:
        g.render(abi.render_params(spp=2, rng_mode=7))
    assert e.value.code == abi.MSK_ERR_INVALID_ARG
    g.close()


def test_a_group_context_shards_this_mode_by_blocks(abi, hostmirror, oracle, golden_lookup):
    """msk_gpu_init(ids = {0, 0}) + MSK_RNG_PCG_BLOCK: the members take every other spiral block (their streams are independent),
    not every other sample (which would draw the same numbers twice: round 4's advisor finding).  The summed film is the
    unsharded one: bit for bit wherever one block contributes, re-associated sums of at most four terms on the block borders;
    a caller's own block shard composes with the members'."""
    flat = hostmirror.cbox_scene(96, 64, coeff_lookup=golden_lookup)
    prm = abi.render_params(spp=3, seed=2, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    o = oracle.scene(flat)
    ref, rst = o.render(prm, threads=8)
    with abi.Context((0, 0)) as grp:
        s = abi.Scene(grp, flat)
        film, st = s.render(prm)
        part = [s.render(abi.render_params(spp=3, seed=2, rng_mode=abi.MSK_RNG_PCG_BLOCK, block_first=k, block_stride=2))[0] for k in (0, 1)]
        with pytest.raises(abi.MskError) as e:
            s.render(abi.render_params(spp=4, seed=2, rng_mode=abi.MSK_RNG_PCG_BLOCK, sample_first=1, sample_stride=2))
        assert e.value.code == abi.MSK_ERR_UNSUPPORTED
        s.close()
    # (shadow rays are counted differently on the two sides: the scalar loop traces one whenever the emitter sample's pdf is not 0,
    # scene.cpp:90-97, the device only when its contribution is not 0)
    assert (st.samples, st.segments) == (rst.samples, rst.segments)
    assert np.allclose(film, ref, rtol=1e-6, atol=1e-6)
    inner = np.zeros((64, 96), bool)
    for by in range(2):
        for bx in range(3):
            inner[by * 32 + 2:by * 32 + 30, bx * 32 + 2:bx * 32 + 30] = True      # pixels only their own block reaches (border 2)
    assert np.array_equal(film[inner].view(np.uint32), ref[inner].view(np.uint32))
    assert np.allclose(part[0] + part[1], ref, rtol=1e-6, atol=1e-6)
    o.close()
