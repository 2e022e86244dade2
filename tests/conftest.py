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
