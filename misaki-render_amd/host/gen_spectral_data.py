#!/usr/bin/env python3
"""gen_spectral_data.py <cie_tables.json> <out.cpp>: the host library's spectral tables as C++ arrays —
float for the renderer (core/spectrum.h:78-80, spectra/d65.cpp:12-27), double for the rgb2spec optimiser
(the decimal values of the public CIE data parsed at double precision, not floats widened)."""
import json
import sys

d = json.load(open(sys.argv[1]))
cie = d["cie1931_x"] + d["cie1931_y"] + d["cie1931_z"]
f32 = lambda a: ", ".join(repr(float(x)) + "f" for x in a)
f64 = lambda a: ", ".join(repr(float(x)) for x in a)
open(sys.argv[2], "w").write(
    "// generated from misaki-render_amd/data/cie_tables.json (public CIE 1931 2-deg observer + D65 data)\n"
    "#include <misaki/render.h>\nnamespace misaki {\n"
    f"static const float k_cie[285] = {{{f32(cie)}}};\n"
    f"static const float k_d65[95] = {{{f32(d['d65'])}}};\n"
    f"static const double k_cie_f64[285] = {{{f64(cie)}}};\n"
    f"static const double k_d65_f64[95] = {{{f64(d['d65'])}}};\n"
    "const float *cie1931_xyz_table() { return k_cie; }\nconst float *d65_table() { return k_d65; }\n"
    "const double *cie1931_xyz_table_f64() { return k_cie_f64; }\nconst double *d65_table_f64() { return k_d65_f64; }\n}\n")
