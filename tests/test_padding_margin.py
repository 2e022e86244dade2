"""The padding rule of the traversal (DESIGN.md §3 rule 4, oracle deviation D10): node boxes are padded by 1e-5, triangle bounds by
0.5e-5 of the scene's SCALE = max(diagonal, largest |coordinate|), so that a conservative tree can never cull a hit the triangle test
accepts and every tree gives the brute-force answer.  This file pins how far that rule is from failing: the oracle's tree and its
brute force still agree on adversarial rays with a TENTH of the padding (the slab tests' rounding errors sit two more orders of
magnitude below), also for a scene translated 1e5 units away from the origin — where a padding tied to the scene's extent alone
(rounds 1-3: 1e-4 of the diagonal) was smaller than an ulp of the coordinates."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_rays", os.path.join(ROOT, "tools", "fuzz_rays.py"))
    fr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fr)
    return fr


def _disagreements(orc, fr, seed, offset, n_rays=12000, scale=1.0):
    rng = np.random.RandomState(seed)
    flat = fr.fz.random_scene(rng)
    if scale != 1.0:
        flat.vertices[:, :3] *= np.float32(scale)
    if offset:
        flat.vertices[:, :3] += np.float32(offset)
    d = flat.desc
    tris = np.array([[flat.vertices[md.first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes) for md in [d.meshes[m]]
                     for f in range(md.first_face, md.first_face + md.face_count)], np.float32)
    rays = fr.adversarial_rays(rng, tris, n_rays)
    brute, tree = orc.scene(flat), orc.scene(flat)
    brute.set_bvh(0)
    tree.set_bvh(1)
    dc = (brute.trace_closest(rays).view(np.uint32) != tree.trace_closest(rays).view(np.uint32)).any(-1)
    da = brute.trace_any(rays) != tree.trace_any(rays)
    brute.close()
    tree.close()
    return int(dc.sum() + da.sum())


@pytest.mark.parametrize("offset", [0.0, 1e5])
@pytest.mark.parametrize("scale", [None, "1e-6"])
def test_tree_equals_brute_force_with_a_tenth_of_the_padding(monkeypatch, scale, offset):
    if scale is None:
        monkeypatch.delenv("MSK_PAD_SCALE", raising=False)
    else:
        monkeypatch.setenv("MSK_PAD_SCALE", scale)
    orc, fr = oracle_binding.load(), _fuzz()
    assert sum(_disagreements(orc, fr, seed, offset) for seed in (0, 6, 7)) == 0       # (seed 6 has a 4 k-triangle mesh)


@pytest.mark.parametrize("scale", [1e-3, 1e3])
def test_the_rule_follows_the_scenes_scale(monkeypatch, scale):
    """The same scenes a thousand times smaller / larger: the padding is relative to the scene's scale, so nothing changes
    (at a tenth of the padding, as above)."""
    monkeypatch.setenv("MSK_PAD_SCALE", "1e-6")
    orc, fr = oracle_binding.load(), _fuzz()
    assert sum(_disagreements(orc, fr, seed, 0.0, scale=scale) for seed in (0, 4, 7)) == 0


def test_the_padding_is_what_keeps_them_equal(monkeypatch):
    """... and without it (1e-9 of the scale: below the rounding errors) they do disagree: the test above tests something."""
    monkeypatch.setenv("MSK_PAD_SCALE", "1e-9")
    orc, fr = oracle_binding.load(), _fuzz()
    assert sum(_disagreements(orc, fr, seed, 0.0) for seed in (0, 4, 7)) > 0
