"""Randomised parity: triangle soups with slivers, every BSDF type with random parameters, one to three area emitters,
optional environment, random integrator properties and block sizes (tools/fuzz_parity.py; 4300 such scenes were run
bit-identical during development) — a slice of it as a regression test."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_scenes_are_valid_oracle_inputs(oracle, abi):
    """CPU: the generator produces scenes the oracle renders to finite films, deterministically per seed."""
    fz = _fuzz()
    for s in (1, 2, 3):
        flat = fz.random_scene(np.random.RandomState(s))
        sc = oracle.scene(flat)
        a, _ = sc.render(abi.render_params(spp=2, seed=s), threads=2)
        b, _ = sc.render(abi.render_params(spp=2, seed=s), threads=3)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.isfinite(a[..., 4]).all() and a[..., 4].min() > 0
        sc.close()


def test_hit_acceptance_does_not_depend_on_the_tree(oracle):
    """Oracle deviation D10, found by this sweep (seed 20657): a shadow ray lying in the plane of a sliver emitter triangle
    gets a numerical Moeller-Trumbore "hit" 40 units outside the triangle (float64: u = -0.13), which a tree reports or
    culls depending on its leaves.  With the bounds predicate brute force and the BVH agree — on this ray and on random
    rays through random sliver soups."""
    import json
    fz = _fuzz()
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "d10_scene.json")))
    soup = fz.hm.MeshSpec("soup", [tuple(tuple(v) for v in t) for t in fixture["triangles"]], (0.5, 0.5, 0.5), radiance=(1, 1, 1))
    sc = oracle.scene(fz.hm.flatten([soup], 16, 16))
    ray = np.array([fixture["ray"]], np.float32)
    # without the bounds predicate the Moeller-Trumbore arithmetic alone reports t = 204.8 on triangle 0 (checked when the
    # fixture was made); with it nothing is hit, with or without the tree
    with_bvh = (sc.trace_any(ray).copy(), sc.trace_closest(ray).copy())
    sc.set_bvh(0)
    assert np.array_equal(sc.trace_any(ray), with_bvh[0]) and with_bvh[0][0] == 0
    assert np.array_equal(sc.trace_closest(ray).view(np.uint32), with_bvh[1].view(np.uint32))
    sc.close()
    rng = np.random.RandomState(7)
    for s in (20657, 11, 12, 13):
        sc = oracle.scene(fz.random_scene(np.random.RandomState(s)))
        o = rng.uniform(-600, 900, (4000, 3)); d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        # a third of the rays lie (almost) in the plane z = const through their origin: the near-parallel regime
        d[::3, 2] *= 1e-7
        rays = np.concatenate([o, np.full((4000, 1), 1e-3), d, np.full((4000, 1), 3000.0)], 1).astype(np.float32)
        sc.set_bvh(1)
        a = sc.trace_closest(rays).copy(); occ = sc.trace_any(rays).copy()
        sc.set_bvh(0)
        assert np.array_equal(sc.trace_closest(rays).view(np.uint32), a.view(np.uint32)) and np.array_equal(sc.trace_any(rays), occ)
        sc.close()


@pytest.mark.gpu
def test_gpu_equals_oracle_on_random_scenes(gpu_ctx, oracle):
    assert _fuzz().sweep(gpu_ctx, oracle, list(range(5000, 5060)) + [20657]) == []


@pytest.mark.gpu
def test_gpu_equals_oracle_on_random_scenes_with_many_samples(gpu_ctx, oracle):
    """The same random scenes at 400 spp: enough samples (>= 1024 regions) for the wavefront loop to run as four loops on
    four streams with every material, emitter and shard kind the generator makes."""
    assert _fuzz().sweep(gpu_ctx, oracle, list(range(7000, 7008)), verbose=False, spp=400) == []


def _fuzz_rays():
    spec = importlib.util.spec_from_file_location("fuzz_rays", os.path.join(ROOT, "tools", "fuzz_rays.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_adversarial_rays_gpu_equals_oracle_brute_force(gpu_ctx, oracle, abi):
    """Rays from vertices / edges / planes of the triangles towards other vertices, along edges, axis-aligned, with tmax
    exactly at the target (tools/fuzz_rays.py; 5 M such rays identical in development, after it had found that the GPU
    dropped first hits whose t exceeds tmax by an ulp): whatever tree the GPU uses vs the oracle's brute force."""
    fr = _fuzz_rays()
    for s in (3, 4, 5, 143):
        rng = np.random.RandomState(s)
        flat = fr.fz.random_scene(rng)
        d = flat.desc
        tris = np.array([[flat.vertices[d.meshes[m].first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes)
                         for f in range(d.meshes[m].first_face, d.meshes[m].first_face + d.meshes[m].face_count)], np.float32)
        rays = fr.adversarial_rays(rng, tris, 20000)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        o.set_bvh(0)
        assert np.array_equal(g.trace_closest(rays).view(np.uint32), o.trace_closest(rays).view(np.uint32)), s
        assert np.array_equal(g.trace_any(rays), o.trace_any(rays)), s
        g.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [1e5, -3e5])
def test_adversarial_rays_on_a_scene_far_from_the_origin(gpu_ctx, oracle, abi, offset):
    """The same sweep on scenes translated away from the origin, where an ulp of a coordinate (0.008 at 1e5) is no longer small
    against the scene's extent: the padding of the boxes and of the D10 bounds follows the scene's SCALE = max(diagonal, largest
    |coordinate|) (tests/test_padding_margin.py), so the LDS-resident binary tree (seed 3) and the quantised 4-wide tree in HBM
    (seed 4: 4.5 k triangles) still give the brute-force answer bit for bit."""
    fr = _fuzz_rays()
    for s in (3, 4):
        rng = np.random.RandomState(s)
        flat = fr.fz.random_scene(rng)
        flat.vertices[:, :3] += np.float32(offset)
        d = flat.desc
        tris = np.array([[flat.vertices[d.meshes[m].first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes)
                         for f in range(d.meshes[m].first_face, d.meshes[m].first_face + d.meshes[m].face_count)], np.float32)
        rays = fr.adversarial_rays(rng, tris, 20000)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        o.set_bvh(0)
        assert np.array_equal(g.trace_closest(rays).view(np.uint32), o.trace_closest(rays).view(np.uint32)), s
        assert np.array_equal(g.trace_any(rays), o.trace_any(rays)), s
        g.close()
        o.close()


@pytest.mark.gpu
def test_gpu_trees_equal_brute_force_with_a_tenth_of_the_padding(gpu_ctx, oracle, abi, monkeypatch):
    """The margin of the padding rule for the DEVICE's slab arithmetic (v_rcp_f32 reciprocals, `plane * idir - o * idir`, the
    quantised nodes' decode): with MSK_PAD_SCALE=1e-6 on both sides — a tenth of the rule's padding — the LDS-resident binary tree
    (seeds 3, 5), the quantised 4-wide tree in HBM (seeds 0, 4) and the same scenes 1e5 units from the origin still give the oracle's
    brute-force answer on adversarial rays, bit for bit (the CPU counterpart: tests/test_padding_margin.py)."""
    monkeypatch.setenv("MSK_PAD_SCALE", "1e-6")
    fr = _fuzz_rays()
    for s, offset in ((3, 0.0), (0, 0.0), (4, 0.0), (5, 1e5), (4, 1e5)):
        rng = np.random.RandomState(s)
        flat = fr.fz.random_scene(rng)
        if offset:
            flat.vertices[:, :3] += np.float32(offset)
        d = flat.desc
        tris = np.array([[flat.vertices[d.meshes[m].first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes)
                         for f in range(d.meshes[m].first_face, d.meshes[m].first_face + d.meshes[m].face_count)], np.float32)
        rays = fr.adversarial_rays(rng, tris, 20000)
        g, o = abi.Scene(gpu_ctx, flat), oracle.scene(flat)
        o.set_bvh(0)
        assert np.array_equal(g.trace_closest(rays).view(np.uint32), o.trace_closest(rays).view(np.uint32)), (s, offset)
        assert np.array_equal(g.trace_any(rays), o.trace_any(rays)), (s, offset)
        g.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bad", ["0", "abc", "-1e-5", "nan", "1", "1e-9", ""])
def test_a_padding_knob_that_is_not_a_sane_number_is_refused(gpu_ctx, abi, hostmirror, golden_lookup, monkeypatch, bad):
    """MSK_PAD_SCALE is honoured for the margin tests only within [1e-7, 1e-3]: zero, garbage or a stray value would silently take
    the padding out of the GPU library AND the oracle at once — the parity tests would still pass (round 4's advisor finding)."""
    monkeypatch.setenv("MSK_PAD_SCALE", bad)
    with pytest.raises(abi.MskError) as e:
        abi.Scene(gpu_ctx, hostmirror.cbox_scene(16, 16, coeff_lookup=golden_lookup))
    assert e.value.code == abi.MSK_ERR_INVALID_ARG and "MSK_PAD_SCALE" in str(e.value)
