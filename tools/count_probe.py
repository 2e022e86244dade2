"""Traversal work per ray on the mesh configs, from an instrumented build (tools/build_variant.sh count -DMSK_COUNT):
MSK_GPU_LIB=gpurun_scratch/libmsk_gpu_count.so python3 tools/count_probe.py c3|c5 [spp]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
which = sys.argv[1] if len(sys.argv) > 1 else "c5"; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
flat = hm.bunny_class_scene(1024) if which == "c3" else hm.teapot_class_scene(1024)
ctx = abi.Context(0); sc = abi.Scene(ctx, flat)
lib = abi.load_library()
buf = (C.c_ulonglong * 16)()
lib.msk_gpu_debug_counts(buf)
film, st = sc.render(abi.render_params(spp=spp))
lib.msk_gpu_debug_counts(buf)
c = list(buf)
rays = max(c[0], 1)
print(f"{which} spp {spp} env WIDE={os.environ.get('MSK_WIDE_BVH', '4')} QUANTUM={os.environ.get('MSK_TRACE_QUANTUM', '4')}: rays {c[0]} (stats: {st.segments + st.shadow_rays}), trace {st.ms_trace:.1f} ms")
print(f"  per ray: inner-node visits {c[4] / rays:.2f}, leaf visits {c[7] / rays:.2f}, triangle tests {c[6] / rays:.2f}, quanta {c[2] / rays:.2f}")
print(f"  wave level: quanta {c[1]}, lanes active per quantum {c[2] / max(c[1], 1):.1f}; inner steps {c[3]} at {c[4] / max(c[3], 1):.1f} lanes; triangle steps {c[5]} at {c[6] / max(c[5], 1):.1f} lanes")
