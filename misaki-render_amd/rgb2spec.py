"""RGB -> spectrum upsampling for the scripting-side host mirror (setup only, once per texture at scene load).

`srgb_model_fetch` is the reference's (src/librender/srgb.cpp:11-28 -> ext/rgb2spec/rgb2spec.c:77-119): a trilinear
fetch, in fp32 and in the reference's operation order, from the res-64 sRGB coefficient table of Jakob & Hanika (2019).
The table is the file the C++ host library uses (misaki-render_amd/lib/srgb.coeff, layout of the reference's
data/srgb.coeff: "SPEC", uint32 res, float scale[res], float data[3][res][res][res][3]).  It is a build artefact:
`make -C misaki-render_amd/host` (what __graft_entry__.build() runs) computes it with the library's own optimiser
(host/src/rgb2spec_table.cpp); if it is missing here, that optimiser is called through the C API and the file is written.
Nothing under oracle/ and nothing of the reference checkout is read.

Deviation (same as the C++ side): pure black returns (0, 0, -inf) — the constant-zero spectrum of render/srgb.h:13-14 —
where rgb2spec_fetch computes 0 * inf and returns NaNs.
"""
import functools
import math
import os
import struct

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
TABLE_PATH = os.environ.get("MSK_SRGB_COEFF") or os.path.join(_PKG, "lib", "srgb.coeff")
F = np.float32


def read_table(path):
    """-> (res, scale float32[res], data float32[3, res, res, res, 3])"""
    with open(path, "rb") as f:
        if f.read(4) != b"SPEC":
            raise ValueError(f"{path}: not an rgb2spec coefficient file")
        (res,) = struct.unpack("<I", f.read(4))
        scale = np.frombuffer(f.read(4 * res), np.float32)
        data = np.frombuffer(f.read(4 * 9 * res ** 3), np.float32)
    if scale.size != res or data.size != 9 * res ** 3:
        raise ValueError(f"{path}: truncated")
    return res, scale, data.reshape(3, res, res, res, 3)


@functools.lru_cache(None)
def table():
    if not os.path.exists(TABLE_PATH):
        if "MSK_SRGB_COEFF" in os.environ:
            raise FileNotFoundError(TABLE_PATH)
        from . import hostlib                       # the C++ host library's optimiser (a few seconds on the host's cores)
        hostlib.rgb2spec_build(64, TABLE_PATH)
    return read_table(TABLE_PATH)


def fetch(res, scale, data, rgb):
    """rgb2spec.c:77-119 in numpy float32 scalars: same products, same sums, same order."""
    c = [max(min(F(x), F(1)), F(0)) for x in rgb]
    l = 0
    for j in (1, 2):
        if c[j] >= c[l]:
            l = j
    z = c[l]
    s = F(res - 1) / z
    x, y = c[(l + 1) % 3] * s, c[(l + 2) % 3] * s
    xi, yi = min(int(x), res - 2), min(int(y), res - 2)
    zi, n = 0, res - 2
    while n > 0:
        half = n >> 1
        mid = zi + half + 1
        if scale[mid] <= z:
            zi, n = mid, n - (half + 1)
        else:
            n = half
    zi = min(zi, res - 2)
    x1, y1 = x - F(xi), y - F(yi)
    x0, y0 = F(1) - x1, F(1) - y1
    z1 = (z - scale[zi]) / (scale[zi + 1] - scale[zi])
    z0 = F(1) - z1
    d = data[l]
    a, b = d[zi], d[zi + 1]
    out = (((a[yi, xi] * x0 + a[yi, xi + 1] * x1) * y0 + (a[yi + 1, xi] * x0 + a[yi + 1, xi + 1] * x1) * y1) * z0 +
           ((b[yi, xi] * x0 + b[yi, xi + 1] * x1) * y0 + (b[yi + 1, xi] * x0 + b[yi + 1, xi + 1] * x1) * y1) * z1)
    return tuple(float(v) for v in out.astype(np.float32))


@functools.lru_cache(4096)
def _fetch_cached(rgb):
    if not all(math.isfinite(v) for v in rgb):
        raise ValueError("srgb_model_fetch: colour is not finite")
    if all(max(min(v, 1.0), 0.0) == 0.0 for v in rgb):
        return (0.0, 0.0, -math.inf)
    res, scale, data = table()
    with np.errstate(all="ignore"):
        return fetch(res, scale, data, rgb)


def srgb_model_fetch(rgb):
    """rgb (3 floats) -> (c0, c1, c2) float32 for S(l) = 1/2 + x/(2 sqrt(1+x^2)), x = (c0 l + c1) l + c2,
    l in nanometres (render/srgb.h:8-19)."""
    return _fetch_cached(tuple(float(F(x)) for x in rgb))


def eval_spectrum(coeff, lam):
    """float64 evaluation of the model (for tests and tools)."""
    lam = np.asarray(lam, np.float64)
    if math.isinf(coeff[2]):
        return np.full(lam.shape, 0.5 + 0.5 * math.copysign(1.0, coeff[2]))
    x = (coeff[0] * lam + coeff[1]) * lam + coeff[2]
    return 0.5 + 0.5 * x / np.sqrt(1.0 + x * x)
