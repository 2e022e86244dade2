#!/usr/bin/env python3
"""Writes tests/golden/oracle_film.json: a regression anchor for the ORACLE itself (the GPU parity tests compare against
the oracle, so a change that moved both sides together would otherwise go unnoticed).  Not reference output — the
reference cannot be built here; it is this repository's oracle at a fixed commit, on three small scenes,
as a SHA-256 of the film bytes plus a few pixels in hex."""
import hashlib
import importlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scenes(hm, lookup):
    glass = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
    # every colour below is one of the reference-fetched triplets of rgb2spec_triplets.json (no solver in the loop)
    gold = {"type": "roughconductor", "alpha": 0.2, "eta": (0.885809, 0.698859, 0.666422), "k": (0.45, 0.30, 0.90), "twosided": True}
    yield "cbox_diffuse", hm.flatten(hm.cbox_meshes(), 48, 40, coeff_lookup=lookup), dict(spp=4, seed=1)
    m = hm.cbox_meshes()
    m[6].bsdf, m[7].bsdf = gold, glass
    yield "cbox_metal_glass", hm.flatten(m, 40, 40, coeff_lookup=lookup), dict(spp=4, seed=2)
    m = hm.cbox_meshes()
    del m[3]
    yield "open_box_environment", hm.flatten(m, 40, 32, coeff_lookup=lookup, env={"radiance": None}), dict(spp=4, seed=3, rr_depth=2)


def main():
    abi = importlib.import_module("misaki-render_amd.abi")
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    import oracle_binding
    orc = oracle_binding.load()
    trip = json.load(open(os.path.join(HERE, "rgb2spec_triplets.json")))
    table = {tuple(round(c, 6) for c in v["rgb"]): tuple(struct.unpack(">f", bytes.fromhex(h))[0] for h in v["coeff_hex"])
             for v in trip.values()}

    def lookup(rgb):
        return table[tuple(round(c, 6) for c in rgb)]
    out = {}
    for name, flat, kw in scenes(hm, lookup):
        sc = orc.scene(flat)
        film, st = sc.render(abi.render_params(**kw), threads=4)
        h, w = film.shape[:2]
        out[name] = {"params": kw, "shape": list(film.shape), "sha256": hashlib.sha256(film.tobytes()).hexdigest(),
                     "samples": int(st.samples), "segments": int(st.segments),
                     "pixels_hex": {f"{y},{x}": film[y, x].tobytes().hex() for y, x in ((h // 2, w // 2), (3, 5), (h - 2, w - 3))}}
        sc.close()
    json.dump(out, open(os.path.join(HERE, "oracle_film.json"), "w"), indent=1)
    print(json.dumps({k: v["sha256"][:16] for k, v in out.items()}))


if __name__ == "__main__":
    main()
