import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress (a kernel that never drains, a reset of the host's GPUs by another tenant) ends after
    seven minutes with every thread's stack on stderr instead of sitting there until the caller's limit (pytest-timeout, when it
    is installed; method "thread": a main thread blocked inside a HIP call never sees a signal).  The slowest GPU test takes 15 s."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(420, method="thread"))


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("misaki-render_amd")


@pytest.fixture(scope="session")
def abi():
    return importlib.import_module("misaki-render_amd.abi")


@pytest.fixture(scope="session")
def hostmirror():
    return importlib.import_module("misaki-render_amd.hostmirror")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def golden():
    import json
    g = os.path.join(ROOT, "tests", "golden")
    return {"triplets": json.load(open(os.path.join(g, "rgb2spec_triplets.json"))),
            "sweep": json.load(open(os.path.join(g, "rgb2spec_sweep.json"))),
            "spectral": json.load(open(os.path.join(g, "spectral_tables.json")))}


@pytest.fixture(scope="session")
def golden_lookup(golden):
    """The PRODUCT's srgb_model_fetch (misaki-render_amd/rgb2spec.py), checked on the way: for every colour the reference's
    own rgb2spec_fetch was recorded for (tests/golden/rgb2spec_triplets.json) the coefficients must be bit-identical.
    Scenes in the tests are therefore built with the product's upsampling, not with injected reference values."""
    import struct
    r2s = importlib.import_module("misaki-render_amd.rgb2spec")
    table = {}
    for v in golden["triplets"].values():
        key = tuple(round(float(x), 7) for x in v["rgb"])
        table[key] = tuple(struct.unpack(">f", bytes.fromhex(h))[0] for h in v["coeff_hex"])

    def lookup(rgb):
        got = r2s.srgb_model_fetch(rgb)
        want = table.get(tuple(round(float(x), 7) for x in rgb))
        assert want is None or got == want, (rgb, got, want)
        return got
    return lookup


@pytest.fixture(scope="session")
def gpu_ctx(abi):
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
