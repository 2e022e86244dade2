"""Adversarial ray sweep for the traversal kernels: rays that start on triangles' vertices / edges / planes and aim at
other vertices, edge points or along triangle planes (the numerically degenerate cases), on random soups with slivers and
lattice-aligned quads.  GPU (whatever tree the scene gets) vs the oracle's BRUTE FORCE, bit for bit.
usage: fuzz_rays.py n_scenes [first_seed]"""
import importlib, importlib.util, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
abi, hm = fz.abi, fz.hm
import oracle_binding


def adversarial_rays(rng, tris, n):
    """tris: float32 [T,3,3]"""
    T = len(tris)
    def point(kind):
        t = tris[rng.randint(0, T)].astype(np.float64)
        if kind == 0: return t[rng.randint(0, 3)]                                   # a vertex
        if kind == 1: a = rng.uniform(); i = rng.randint(0, 3); return t[i] * a + t[(i + 1) % 3] * (1 - a)   # on an edge
        b = rng.dirichlet([1, 1, 1]); return b @ t                                   # inside
    rays = np.zeros((n, 8), np.float32)
    for k in range(n):
        o = point(rng.randint(0, 3)); tgt = point(rng.randint(0, 3))
        mode = rng.randint(0, 5)
        if mode == 0:                      # in-plane direction of the origin's... any triangle: edge direction
            t = tris[rng.randint(0, T)].astype(np.float64); d = t[1] - t[0] + (t[2] - t[0]) * rng.choice([0.0, rng.uniform(-1, 1)])
        elif mode == 1:                    # axis-aligned
            d = np.zeros(3); d[rng.randint(0, 3)] = rng.choice([-1.0, 1.0])
        else:
            d = tgt - o
        if not np.any(d): d = np.array([1.0, 0, 0])
        d = d / np.linalg.norm(d)
        back = rng.choice([0.0, 0.0, rng.uniform(1, 300)])       # start behind the point so that the point itself is a hit candidate
        o = o - d * back
        rays[k] = [o[0], o[1], o[2], rng.choice([0.0, 1e-4, 0.05]), d[0], d[1], d[2], rng.choice([np.inf, 5000.0, np.linalg.norm(tgt - o) * rng.choice([1.0, 0.9991])])]
    return rays


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = abi.Context(0); orc = oracle_binding.load()
    bad = 0; total = 0
    for s in range(seed0, seed0 + n):
        rng = np.random.RandomState(s)
        flat = fz.random_scene(rng)
        d = flat.desc
        tris = np.array([[flat.vertices[md.first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes) for md in [d.meshes[m]]
                         for f in range(md.first_face, md.first_face + md.face_count)], np.float32)
        rays = adversarial_rays(rng, tris, 20000)
        g, o = abi.Scene(ctx, flat), orc.scene(flat)
        o.set_bvh(0)
        hit_g, hit_o = g.trace_closest(rays), o.trace_closest(rays)
        any_g, any_o = g.trace_any(rays), o.trace_any(rays)
        dc = (hit_g.view(np.uint32) != hit_o.view(np.uint32)).any(-1); da = any_g != any_o
        total += len(rays)
        if dc.any() or da.any():
            bad += 1
            k = int(np.argmax(dc | da))
            print("seed %d: closest differs on %d rays, any-hit on %d; e.g. ray %s gpu %s oracle %s" % (s, int(dc.sum()), int(da.sum()), rays[k].tolist(), hit_g[k], hit_o[k]))
        g.close(); o.close()
    print("ray fuzz: %d scenes, %d rays, %d scenes with a difference" % (n, total, bad))
    sys.exit(1 if bad else 0)
