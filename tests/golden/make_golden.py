#!/usr/bin/env python3
"""Regenerates the DATA fixtures under tests/golden/ from the reference checkout.

Run in the build container only (needs /root/reference and `make -C oracle ref`).
Outputs are numbers, not source text:

  spectral_tables.json   CIE 1931 2-degree observer (95 samples, 360..830 nm) and the
                         D65 illuminant table the reference's host side holds
                         (src/librender/spectrum.cpp:8-111, spectra/d65.cpp:12-27)
  rgb2spec_triplets.json rgb -> (c0,c1,c2) returned by the reference's own
                         rgb2spec_fetch (ext/rgb2spec/rgb2spec.c:77-119, compiled into
                         oracle/_ref/librgb2spec.so) on the res-64 table its own optimiser
                         wrote (oracle/_ref/srgb.coeff), as float32 bit patterns
  rgb2spec_sweep.json    the same for 640 more colours (seeded random in the cube, near-black,
                         greys, primaries / secondaries, table nodes and cell boundaries,
                         out-of-gamut inputs that the fetch clamps): rgb and coefficients as
                         float32 bit patterns, plus the sha256 of the reference-built table
"""
import ctypes as C
import json
import os
import re
import struct

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def floats_in(path, start_marker, end_marker="};"):
    txt = open(path).read()
    body = txt[txt.index(start_marker):]
    body = body[body.index("{") + 1:body.index(end_marker)]
    return [float(x) for x in re.findall(r"[-+]?\d+\.\d*(?:[eE][-+]?\d+)?|[-+]?\d+(?=f)", body.replace("float(", "("))]


def f32_hex(x):
    return struct.pack(">f", x).hex()


def main():
    cie = floats_in(f"{REF}/src/librender/spectrum.cpp", "cie1931_tbl")
    assert len(cie) == 95 * 3, len(cie)
    d65 = floats_in(f"{REF}/src/librender/spectra/d65.cpp", "d65_data")
    assert len(d65) == 95, len(d65)
    json.dump({"cie1931_xyz": cie, "d65": d65, "lambda_min": 360.0, "lambda_max": 830.0, "samples": 95},
              open(f"{HERE}/spectral_tables.json", "w"), indent=0)

    lib = C.CDLL(f"{ROOT}/oracle/_ref/librgb2spec.so")
    lib.rgb2spec_load.restype = C.c_void_p
    lib.rgb2spec_load.argtypes = [C.c_char_p]
    lib.rgb2spec_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    model = lib.rgb2spec_load(f"{ROOT}/oracle/_ref/srgb.coeff".encode())
    assert model
    colors = {
        "luminaire_reflectance": (0.936461, 0.740433, 0.705267),
        "white": (0.885809, 0.698859, 0.666422),
        "green": (0.105421, 0.37798, 0.076425),
        "red": (0.570068, 0.0430135, 0.0443706),
        "box": (0.45, 0.30, 0.90),
        "emitter_40_normalised": (0.5, 0.5, 0.5),   # srgb_d65.cpp:18-22: 40 / (2*max) = 0.5
        "grey_0.5": (0.5, 0.5, 0.5),
        "white_1": (1.0, 1.0, 1.0),
        "primary_r": (1.0, 0.0, 0.0),
        "mid_1": (0.2, 0.7, 0.4),
        "mid_2": (0.8, 0.8, 0.1),
        "mid_3": (0.05, 0.15, 0.6),
    }
    out = {}
    for name, rgb in colors.items():
        a = (C.c_float * 3)(*rgb)
        o = (C.c_float * 3)()
        lib.rgb2spec_fetch(model, a, o)
        out[name] = {"rgb": list(rgb), "coeff_hex": [f32_hex(o[i]) for i in range(3)],
                     "coeff": [float(o[i]) for i in range(3)]}
    json.dump(out, open(f"{HERE}/rgb2spec_triplets.json", "w"), indent=1)

    import hashlib
    import numpy as np
    rng = np.random.default_rng(20261004)
    cols = [rng.random(3) for _ in range(400)]
    cols += [rng.random(3) * 10.0 ** -rng.integers(1, 7) for _ in range(60)]                 # dark colours
    cols += [np.full(3, v) for v in np.linspace(0.0, 1.0, 41)[1:]]                           # greys (black excluded: NaN there)
    cols += [np.array(c, float) for c in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 1))]
    cols += [np.array(c, float) * v for c in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1)) for v in (0.25, 0.5, 0.75)]
    scale = np.fromfile(f"{ROOT}/oracle/_ref/srgb.coeff", np.float32, 64, offset=8)
    for k in (1, 5, 12, 13, 31, 62, 63):                                                     # table nodes and cell faces
        for i, j in ((0, 0), (63, 63), (17, 40), (32, 1)):
            b = float(scale[k])
            cols.append(np.array([b * i / 63.0, b * j / 63.0, b]))
            cols.append(np.array([b, b * i / 63.0, b * j / 63.0]))
    cols += [np.array(c, float) for c in ((2.0, 0.5, 0.25), (0.3, 1.5, -0.2), (-1.0, -2.0, 0.7), (1.0, 1.0, 0.999999), (5e-8, 0.0, 0.0))]
    cols += [rng.random(3) for _ in range(640 - len(cols))]
    sweep = []
    for rgb in cols:
        rgb = np.asarray(rgb, np.float32)
        a = (C.c_float * 3)(*[float(v) for v in rgb])
        o = (C.c_float * 3)()
        lib.rgb2spec_fetch(model, a, o)
        sweep.append([f32_hex(float(v)) for v in rgb] + [f32_hex(o[i]) for i in range(3)])
    sha = hashlib.sha256(open(f"{ROOT}/oracle/_ref/srgb.coeff", "rb").read()).hexdigest()
    json.dump({"_comment": "rgb (3 x float32 hex) -> coefficients (3 x float32 hex) from the reference's rgb2spec_fetch on the "
                           "res-64 sRGB table the reference's rgb2spec_opt wrote in this container (platform libm); table_sha256 "
                           "is that file's hash", "table_sha256": sha, "table_bytes": os.path.getsize(f"{ROOT}/oracle/_ref/srgb.coeff"),
               "rows": sweep}, open(f"{HERE}/rgb2spec_sweep.json", "w"), indent=0)
    print("wrote fixtures:", len(cie), len(d65), len(out), len(sweep))


if __name__ == "__main__":
    main()
