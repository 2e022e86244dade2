"""How much of a bench step is not kernel time: wall time of msk_gpu_render_device with and without statistics (the HIP
events around every launch exist only when the caller asks for msk_stats)."""
import importlib, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
ctx = abi.Context(0); sc = abi.Scene(ctx, hm.cbox_scene(512, 512))
film = torch.zeros((512, 512, 5), dtype=torch.float32, device="cuda")
prm = abi.render_params(spp=512)
for _ in range(2):
    sc.render_device(prm, film.data_ptr())
def wall(with_stats, n=6):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if with_stats:
            st = sc.render_device(prm, film.data_ptr())
        else:
            ctx.check(ctx.lib.msk_gpu_render_device(sc.handle, C.byref(prm), C.c_void_p(film.data_ptr()), C.c_void_p(0), None))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
a = wall(True); b = wall(False); a2 = wall(True); b2 = wall(False)
st = sc.render_device(prm, film.data_ptr())
print("wall ms with stats %.2f / %.2f, without %.2f / %.2f; kernels: trace %.2f shade %.2f resolve %.2f = %.2f" % (
    a, a2, b, b2, st.ms_trace, st.ms_shade, st.ms_resolve, st.ms_trace + st.ms_shade + st.ms_resolve))
