"""RGB -> spectrum upsampling for the host side (setup only, once per texture at scene load).

Own implementation of the Jakob & Hanika (2019) sigmoid-polynomial model that the reference
reaches through `srgb_model_fetch` (src/librender/srgb.cpp:11-28 -> ext/rgb2spec).  The
reference interpolates a precomputed 64^3 coefficient table; this module instead solves for
the three coefficients of the requested colour directly (Gauss-Newton on the CIELAB residual
with a homotopy from mid-grey), so it needs no 9.4 MB data file.  The spectra it returns
integrate to the same sRGB colour under D65 as the reference's; the coefficients themselves
differ from the table's trilinear blend in the 3rd-4th digit (tests/test_host_mirror.py bounds
the spectral difference).  Evaluation on the device follows render/srgb.h:8-19 exactly.
"""
import functools
import math

import numpy as np

_XYZ_TO_SRGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556],
                         [0.055648, -0.204043, 1.057311]])
_SRGB_TO_XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]])


@functools.lru_cache(None)
def _quadrature():
    from .hostmirror import cie_tables
    xyz, d65 = cie_tables()
    xyz = xyz.astype(np.float64).reshape(3, 95)
    d65 = d65.astype(np.float64)
    lam5 = np.linspace(360.0, 830.0, 95)
    lam = np.linspace(360.0, 830.0, 471 * 2 - 1)          # 0.5 nm
    w = np.full(lam.size, lam[1] - lam[0])
    w[0] *= 0.5
    w[-1] *= 0.5
    cmf = np.stack([np.interp(lam, lam5, xyz[i]) for i in range(3)])
    ill = np.interp(lam, lam5, d65)
    ill = ill / np.sum(cmf[1] * ill * w)                   # Y of the illuminant = 1
    rgb_tbl = (_XYZ_TO_SRGB @ cmf) * ill * w               # (3, n): spectrum -> linear sRGB
    white_xyz = (cmf * ill * w).sum(1)
    t = (lam - 360.0) / 470.0
    return t, rgb_tbl, white_xyz


def _sigmoid(x):
    return 0.5 + 0.5 * x / np.sqrt(1.0 + x * x)


def _lab(rgb, white):
    xyz = _SRGB_TO_XYZ @ rgb
    def f(v):
        d = 6.0 / 29.0
        return np.where(v > d ** 3, np.cbrt(np.maximum(v, 0)), v / (3 * d * d) + 4.0 / 29.0)
    fx, fy, fz = f(xyz[0] / white[0]), f(xyz[1] / white[1]), f(xyz[2] / white[2])
    return np.array([116 * fy - 16, 500 * (fx - fy), 200 * (fy - fz)])


def _model_rgb(c, t, tbl):
    return tbl @ _sigmoid((c[0] * t + c[1]) * t + c[2])


def _gauss_newton(c, target_lab, t, tbl, white, iters=20):
    for _ in range(iters):
        r = _lab(_model_rgb(c, t, tbl), white) - target_lab
        if float(r @ r) < 1e-12:
            break
        J = np.empty((3, 3))
        eps = 1e-4
        for i in range(3):
            d = np.zeros(3)
            d[i] = eps
            J[:, i] = (_lab(_model_rgb(c + d, t, tbl), white) - _lab(_model_rgb(c - d, t, tbl), white)) / (2 * eps)
        try:
            step = np.linalg.solve(J, r)
        except np.linalg.LinAlgError:
            break
        c = c - step
        m = np.max(np.abs(c))
        if m > 200:
            c = c * (200 / m)
    return c


@functools.lru_cache(4096)
def _fit(rgb):
    t, tbl, white = _quadrature()
    target = np.clip(np.asarray(rgb, np.float64), 0.0, 1.0)
    if target[0] == target[1] == target[2]:
        v = target[0]
        if v <= 0.0:
            return (0.0, 0.0, -math.inf)
        if v >= 1.0:
            return (0.0, 0.0, math.inf)
        return (0.0, 0.0, (v - 0.5) / math.sqrt(v * (1 - v)))
    c = np.zeros(3)
    grey = np.full(3, 0.5)
    for s in np.linspace(0.0, 1.0, 17)[1:]:
        c = _gauss_newton(c, _lab((1 - s) * grey + s * target, white), t, tbl, white)
    return tuple(float(x) for x in c)


def srgb_model_fetch(rgb):
    """rgb (3 floats in [0,1]) -> (c0, c1, c2) float32 for S(l) = 1/2 + x/(2 sqrt(1+x^2)),
    x = (c0 l + c1) l + c2 with l in nanometres (render/srgb.h:8-19)."""
    A, B, Cc = _fit(tuple(float(np.float32(x)) for x in rgb))
    if math.isinf(Cc):
        return (0.0, 0.0, Cc)
    c0, c1 = 360.0, 1.0 / 470.0
    out = (A * c1 * c1, B * c1 - 2 * A * c0 * c1 * c1, Cc - B * c0 * c1 + A * (c0 * c1) ** 2)
    return tuple(float(np.float32(x)) for x in out)


def eval_spectrum(coeff, lam):
    """float64 evaluation of the model (for tests and tools)."""
    lam = np.asarray(lam, np.float64)
    if math.isinf(coeff[2]):
        return np.full(lam.shape, 0.5 + 0.5 * math.copysign(1.0, coeff[2]))
    return _sigmoid((coeff[0] * lam + coeff[1]) * lam + coeff[2])
