"""Randomised parity sweep: random triangle soups (including slivers and near-degenerate triangles), random BSDFs of
every type, one to three area emitters, optional environment, random integrator properties — the GPU film must equal
the oracle's bit for bit on every one.  usage: fuzz_parity.py n_scenes [first_seed [spp]]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
import oracle_binding


def random_scene(rng):
    meshes = []
    n_mesh = rng.randint(3, 9)
    n_light = rng.randint(1, 4)
    for i in range(n_mesh):
        faces = []
        for _ in range(rng.randint(1, 12)):
            c = rng.uniform(-1, 1, 3) * 200 + np.array([278, 273, 280])
            size = 10 ** rng.uniform(0.3, 2.5)
            kind = rng.randint(0, 12)
            if kind >= 10:     # axis-aligned quad on a lattice (exact zeros in normals and ray/plane arithmetic)
                c = np.round(c / 50) * 50
                ax = rng.randint(0, 3); ext = np.round(rng.uniform(20, 300, 2) / 10) * 10
                e0 = np.zeros(3); e1_ = np.zeros(3); e0[(ax + 1) % 3] = ext[0]; e1_[(ax + 2) % 3] = ext[1]
                pts = [c, c + e0, c + e0 + e1_, c + e1_]
            elif kind == 0:      # sliver
                a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
                pts = [c, c + a, c + a * (1 + 1e-4) + b * 1e-4]
            elif kind == 1:    # quad
                a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
                pts = [c, c + a, c + a + b, c + b]
            else:
                pts = [c + rng.normal(size=3) * size for _ in range(3)]
            faces.append(tuple(tuple(float(np.float32(v)) for v in p) for p in pts))
        refl = tuple(float(x) for x in rng.uniform(0.05, 0.95, 3))
        t = rng.randint(0, 4)
        bsdf = None
        if t == 1:
            bsdf = {"type": "roughconductor", "alpha": float(10 ** rng.uniform(-2.5, -0.3)), "eta": tuple(rng.uniform(0.1, 3, 3)),
                    "k": tuple(rng.uniform(0.5, 4, 3)), "twosided": bool(rng.randint(0, 2)), "sample_visible": bool(rng.randint(0, 2))}
        elif t == 2:
            bsdf = {"type": "roughdielectric", "alpha": float(rng.choice([0.0, 10 ** rng.uniform(-2.5, -0.3)])), "int_ior": float(rng.uniform(1.05, 2.6)),
                    "ext_ior": float(rng.choice([1.0, 1.33])), "sample_visible": bool(rng.randint(0, 2))}
        elif t == 3:
            bsdf = {"type": "roughconductor", "alpha": (float(rng.uniform(0.02, 0.5)), float(rng.uniform(0.02, 0.5))),
                    "eta": (1.5, 1.5, 1.5), "k": (3.0, 3.0, 3.0), "twosided": True}
        if t == 0:
            # textured reflectance on some diffuse surfaces; drawn from a side stream so that the seed -> geometry mapping of
            # earlier sweeps stays what it was
            trng = np.random.RandomState(int(refl[0] * 1e9) % (2 ** 31))
            if trng.randint(0, 6) == 0:
                refl = float(trng.choice([0.0, 1.0, trng.uniform(0.05, 0.95)]))       # a `uniform` spectrum (spectra/uniform.cpp)
            elif trng.randint(0, 3) == 0:
                m = np.zeros(16)
                m[[0, 1, 2, 4, 5, 6]] = trng.uniform(-12, 12, 6) * (trng.uniform(size=6) < 0.8)
                m[15] = 1
                bsdf = {"type": "diffuse", "twosided": bool(trng.randint(0, 2)),
                        "texture": {"type": "checkerboard", "color0": refl, "color1": tuple(float(x) for x in trng.uniform(0.02, 0.98, 3)),
                                    "matrix": [float(np.float32(x)) for x in m]}}
        rad = tuple(float(x) for x in rng.uniform(1, 30, 3)) if i < n_light else None
        meshes.append(hm.MeshSpec("m%d" % i, faces, refl, radiance=rad, bsdf=bsdf))
    if rng.randint(0, 7) == 0:      # a mesh big enough for the HBM traversal kernels (tree > 48 KB)
        res = int(rng.randint(25, 70))
        blob = hm.blob_mesh("blob", tuple(float(x) for x in rng.uniform(150, 400, 3)), float(rng.uniform(40, 120)), res, res,
                            tuple(float(x) for x in rng.uniform(0.1, 0.9, 3)), seed=int(rng.randint(0, 100)), bump=float(rng.uniform(0, 0.3)))
        if rng.randint(0, 2):
            blob.bsdf = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
        meshes.append(blob)
    env = None
    if rng.randint(0, 3) == 0:
        env = {"radiance": tuple(float(x) for x in rng.uniform(0.1, 1.0, 3)) if rng.randint(0, 2) else None, "first": bool(rng.randint(0, 2))}
    camera = None
    if rng.randint(0, 5) == 0:       # the whole scene (and the camera) at another scale: epsilons, clip planes, pdf magnitudes
        sc = float(rng.choice([1e-3, 1e-2, 30.0, 1e3]))
        for m in meshes:
            m.faces = [tuple(tuple(float(np.float32(v * sc)) for v in p) for p in f) for f in m.faces]
        c = hm.CBOX_CAMERA
        camera = dict(fov=float(rng.uniform(20, 90)), near=c["near"] * sc, far=c["far"] * sc, origin=tuple(v * sc for v in c["origin"]),
                      target=tuple(v * sc for v in c["target"]), up=c["up"])
    frng = np.random.RandomState(len(meshes) * 7919 + int(meshes[0].faces[0][0][0] * 1e3) % 100003)    # side stream: filter width
    stddev = float(frng.choice([0.5, 0.5, 0.5, 0.3, 0.625, 0.9]))
    fw, fh = [(48, 40), (48, 40), (33, 17), (64, 64), (7, 5), (100, 3), (1, 1), (37, 53)][frng.randint(0, 8)]      # ragged films, films smaller than a block
    crop = None
    if frng.randint(0, 4) == 0:      # a crop window (film.cpp:12-21): anywhere in the film, down to one pixel (side stream again)
        cw, ch = int(frng.randint(1, fw + 1)), int(frng.randint(1, fh + 1))
        crop = (int(frng.randint(0, fw - cw + 1)), int(frng.randint(0, fh - ch + 1)), cw, ch)
    # tabulated (`regular`) spectra on a quarter of the scenes (ABI v7; side stream): any of a diffuse reflectance, a conductor's
    # eta / k, a dielectric's transmittance, an emitter's radiance, the environment's — tables of 2 … 95 values on grids that
    # cover the sampled wavelengths, start inside them or end inside them (the end segments are then continued)
    trng = np.random.RandomState((len(meshes) * 104729 + int(abs(meshes[0].faces[0][0][1]) * 1e3)) % 1000003)
    if trng.randint(0, 4) == 0:
        def table(lo, hi):
            n = int(trng.choice([2, 3, 5, 16, 31, 48, 95]))
            l0 = float(np.float32(trng.choice([300.0, 360.0, 380.0, 450.0])))
            l1 = float(np.float32(l0 + trng.choice([150.0, 320.0, 470.0, 600.0])))
            return hm.Regular(l0, l1, trng.uniform(lo, hi, n).astype(np.float32))
        for m in meshes:
            if trng.randint(0, 3):
                continue
            if m.bsdf is None and not np.isscalar(m.reflectance):
                m.reflectance = table(0.02, 0.95)
            elif m.bsdf and m.bsdf["type"] == "roughconductor":
                m.bsdf["eta"], m.bsdf["k"] = table(0.1, 3.0), table(0.5, 4.0)
                if trng.randint(0, 2):
                    m.bsdf["specular_reflectance"] = table(0.3, 1.0)
            elif m.bsdf and m.bsdf["type"] == "roughdielectric":
                m.bsdf["specular_transmittance"] = table(0.2, 1.0)
            if m.radiance is not None and trng.randint(0, 2):
                m.radiance = table(0.01, 0.4)
        if env is not None and trng.randint(0, 2):
            env["radiance"] = table(0.001, 0.02)
    flat = hm.flatten(meshes, fw, fh, env=env, camera=camera, filter_stddev=stddev, crop=crop)
    # vertex normals (perturbed face normals) and texture coordinates on some meshes: mesh.cpp:68-96
    verts, faces = flat.vertices, flat.faces
    for i in range(flat.desc.n_meshes):
        md = flat.desc.meshes[i]
        mode = rng.randint(0, 4)
        if mode == 0 or md.face_count == 0:
            continue
        for f in range(md.first_face, md.first_face + md.face_count):
            idx = md.first_vertex + faces[f]
            p = verts[idx, :3].astype(np.float64)
            n = np.cross(p[1] - p[0], p[2] - p[0])
            ln = np.linalg.norm(n)
            n = n / ln if ln > 0 else np.array([0.0, 0.0, 1.0])
            for k in idx:
                q = n + rng.normal(size=3) * 0.3
                verts[k, 3:6] = (q / np.linalg.norm(q)).astype(np.float32)
                verts[k, 6:8] = rng.uniform(0, 1, 2).astype(np.float32)
        md.has_normals = 1 if mode in (1, 3) else 0
        md.has_texcoords = 1 if mode in (2, 3) else 0
    return flat


def random_params(rng):
    kw = dict(spp=int(rng.choice([1, 4, 4, 7])), seed=int(rng.randint(0, 1000)), rr_depth=int(rng.choice([1, 2, 5])),
              max_depth=int(rng.choice([-1, -1, 1, 3, 6])), hide_emitters=int(rng.randint(0, 2)), block_size=int(rng.choice([8, 16, 32])))
    if rng.randint(0, 4) == 0:       # a multi-GPU shard of the job (tiles or sample indices)
        world = int(rng.randint(2, 5))
        if rng.randint(0, 2):
            kw.update(block_first=int(rng.randint(0, world)), block_stride=world)
        elif rng.randint(0, 2):
            kw.update(sample_first=int(rng.randint(0, world)), sample_stride=world)
        else:                        # a contiguous sample range [a, spp)
            kw.update(sample_first=int(rng.randint(0, kw["spp"])), sample_stride=1)
    return kw


def sweep(ctx, orc, seeds, verbose=True, spp=None):
    bad = []
    for s in seeds:
        rng = np.random.RandomState(s)
        flat = random_scene(rng)
        kw = random_params(rng)
        if spp:
            kw["spp"] = spp
        g, o = abi.Scene(ctx, flat), orc.scene(flat)
        film, st = g.render(abi.render_params(**kw))
        ref, rst = o.render(abi.render_params(**kw), threads=8)
        if rng.randint(0, 4) == 0:        # the "aov" integrator on the same scene, random channel list
            types = [int(t) for t in rng.randint(0, 6, rng.randint(1, 5))]
            if types.count(5) > 1:
                types = [t for t in types if t != 5] + [5]
            fa, _ = g.render_aov(abi.render_params(**kw), types)
            ra, _ = o.render_aov(abi.render_params(**kw), types)
            if not np.array_equal(fa.view(np.uint32), ra.view(np.uint32)):
                film = None
        if film is None or not np.array_equal(film.view(np.uint32), ref.view(np.uint32)):
            bad.append(s)
            if verbose:
                print("seed %d differs (%s) params %s" % (s, "aov film" if film is None else "film", kw))
        g.close(); o.close()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else None        # optional: samples per pixel for every scene
    bad = sweep(abi.Context(0), oracle_binding.load(), range(seed0, seed0 + n), spp=spp)
    print("fuzz: %d scenes, %d with a film different from the oracle's" % (n, len(bad)))
    sys.exit(1 if bad else 0)
