"""BASELINE config 1 (cbox 256x256 @ 16 spp) in both sampler modes, on the GPU and on the CPU oracle (8 threads): the device's one-lane-
per-block path of MSK_RNG_PCG_BLOCK (csrc/msk_serial.h) beside the wavefront path, with a bit comparison of the films."""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
import oracle_binding
flat = hm.cbox_scene(256, 256)
ctx = abi.Context(0); g = abi.Scene(ctx, flat); o = oracle_binding.load().scene(flat)
for mode, name in ((abi.MSK_RNG_PCG_BLOCK, "pcg_block"), (abi.MSK_RNG_COUNTER, "counter")):
    prm = abi.render_params(spp=16, rng_mode=mode)
    g.render(prm)
    t0 = time.time(); f, st = g.render(prm); tg = time.time() - t0
    t0 = time.time(); r, rst = o.render(prm, threads=8); tc = time.time() - t0
    import numpy as np
    print(name, "gpu %.1f ms (device %.1f ms)" % (tg * 1e3, st.ms_total), "cpu oracle 8 threads %.1f ms" % (tc * 1e3), "bit-identical", bool(np.array_equal(f, r)), "samples", st.samples)
