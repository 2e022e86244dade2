"""How much does BSDF divergence inside a chunk cost k_shade_gen's general variant?  Shade time per segment for the
Cornell box with (c) every surface diffuse (a dummy conductor triangle outside the view forces the general variant),
(a) every surface a rough conductor, (b) the boxes conductors and the walls diffuse."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
ctx = abi.Context(0)
gold = {"type": "roughconductor", "alpha": 0.2, "eta": (0.143, 0.375, 1.442), "k": (3.983, 2.386, 1.603), "twosided": True}
dummy = hm.MeshSpec("dummy", [((-5000, -5000, -5000), (-5001, -5000, -5000), (-5000, -5001, -5000))], hm.WHITE, bsdf=dict(gold))
def scene(kind):
    m = hm.cbox_meshes()
    for i, mesh in enumerate(m):
        if kind == "all_conductor" and i > 0: mesh.bsdf = dict(gold)
        if kind == "mixed" and i in (6, 7): mesh.bsdf = dict(gold)
        if kind == "mixed_half" and i in (2, 3, 6, 7): mesh.bsdf = dict(gold)
    return hm.flatten(m + [dummy], 512, 512)
for kind in ("all_diffuse", "all_conductor", "mixed", "mixed_half"):
    sc = abi.Scene(ctx, scene(kind))
    sc.render(abi.render_params(spp=16))
    for sort in ("1", "0"):                       # material-sorted shading on / off (read per render)
        os.environ["MSK_SORT"] = sort
        best = None
        for _ in range(3):
            _, st = sc.render(abi.render_params(spp=128))
            if best is None or st.ms_shade < best.ms_shade: best = st
        print("%-14s MSK_SORT=%s shade %.2f ms, %.1f M segments -> %.4f ns/segment; trace %.2f ms; total %.2f ms" % (
            kind, sort, best.ms_shade, best.segments / 1e6, best.ms_shade * 1e6 / best.segments, best.ms_trace, best.ms_total))
    sc.close()
