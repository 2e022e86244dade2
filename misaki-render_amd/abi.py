"""ctypes mirror of include/msk_gpu.h (the C ABI of the MI355X back end).

This is the Python-side binding a maintainer would use from a scripting host;
the structs below must stay field-for-field identical to the header.  The
library is loaded from misaki-render_amd/lib/libmsk_gpu.so and there is NO
fallback: if the HIP extension is missing, loading raises.
"""
import ctypes as C
import operator
import os

import numpy as np

MSK_ABI_VERSION = 7
MSK_REGULAR_MAX = 95
MSK_OK = 0
MSK_ERR_INVALID_ARG, MSK_ERR_NO_DEVICE, MSK_ERR_HIP, MSK_ERR_OOM, MSK_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
MSK_BSDF_DIFFUSE, MSK_BSDF_ROUGHCONDUCTOR, MSK_BSDF_ROUGHDIELECTRIC = 0, 1, 2
MSK_EMITTER_AREA, MSK_EMITTER_CONSTANT = 0, 1
MSK_TEXTURE_CHECKERBOARD = 1
MSK_EMITTER_AREA = 0
MSK_RNG_PCG_BLOCK, MSK_RNG_COUNTER = 0, 1
MSK_CIE_SAMPLES = 95
MSK_FILTER_RESOLUTION = 32


class MeshDesc(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("vertex_count", C.c_uint32),
                ("first_face", C.c_uint32), ("face_count", C.c_uint32),
                ("bsdf_id", C.c_int32), ("emitter_id", C.c_int32),
                ("has_normals", C.c_uint32), ("has_texcoords", C.c_uint32)]


class RegularSpectrumDesc(C.Structure):
    _fields_ = [("lambda_min", C.c_float), ("lambda_max", C.c_float), ("size", C.c_uint32), ("first_value", C.c_uint32)]


class SpectrumDesc(C.Structure):
    _fields_ = [("coeff", C.c_float * 3), ("scale", C.c_float), ("regular", C.c_uint32)]


class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("back_bsdf", C.c_int32), ("reflectance", C.c_float * 3),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("sample_visible", C.c_int32),
                ("eta", SpectrumDesc), ("k", SpectrumDesc), ("specular_reflectance", SpectrumDesc),
                ("specular_transmittance", SpectrumDesc), ("ior_eta", C.c_float), ("ior_inv_eta", C.c_float),
                ("reflectance_texture", C.c_uint32), ("reflectance_scale", C.c_float), ("reflectance_regular", C.c_uint32)]


class TextureDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("color0", C.c_float * 3), ("color1", C.c_float * 3), ("to_uv", C.c_float * 6),
                ("reserved", C.c_float * 3)]


class EmitterDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("mesh_id", C.c_int32), ("radiance", C.c_float * 3),
                ("d65_scale", C.c_float), ("radiance_regular", C.c_uint32)]


class CameraDesc(C.Structure):
    _fields_ = [("sample_to_camera", C.c_float * 16), ("to_world", C.c_float * 16),
                ("near_clip", C.c_float), ("far_clip", C.c_float)]


class FilmDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("filter_radius", C.c_float),
                ("filter_lut", C.c_float * (MSK_FILTER_RESOLUTION + 1)),
                ("crop_offset", C.c_int32 * 2), ("crop_size", C.c_int32 * 2)]     # {0, 0} size = the whole film


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_meshes", C.c_uint32), ("n_bsdfs", C.c_uint32),
                ("n_emitters", C.c_uint32),
                ("meshes", C.POINTER(MeshDesc)), ("bsdfs", C.POINTER(BsdfDesc)),
                ("emitters", C.POINTER(EmitterDesc)),
                ("vertices", C.POINTER(C.c_float)), ("faces", C.POINTER(C.c_uint32)),
                ("n_vertices", C.c_uint32), ("n_faces", C.c_uint32),
                ("camera", CameraDesc), ("film", FilmDesc),
                ("cie1931_xyz", C.POINTER(C.c_float)), ("d65", C.POINTER(C.c_float)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(TextureDesc)),
                ("n_regular_spectra", C.c_uint32), ("n_regular_values", C.c_uint32),
                ("regular_spectra", C.POINTER(RegularSpectrumDesc)), ("regular_values", C.POINTER(C.c_float))]


class RenderParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("seed", C.c_uint64), ("rng_mode", C.c_int32),
                ("rr_depth", C.c_int32), ("max_depth", C.c_int32), ("hide_emitters", C.c_int32),
                ("block_size", C.c_int32), ("block_first", C.c_uint32), ("block_stride", C.c_uint32),
                ("sample_first", C.c_uint32), ("sample_stride", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("iterations", C.c_uint32), ("passes", C.c_uint32), ("ms_total", C.c_float),
                ("ms_generate", C.c_float), ("ms_trace", C.c_float), ("ms_shade", C.c_float),
                ("ms_resolve", C.c_float), ("n_trace_launches", C.c_uint32),
                ("n_shade_launches", C.c_uint32), ("launches_trace", C.c_uint32), ("launches_shade", C.c_uint32),
                ("launches_wavefront", C.c_uint32), ("invalid_samples", C.c_uint64),
                ("bytes_shade", C.c_uint64), ("bytes_trace", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def render_params(spp, seed=0, rng_mode=MSK_RNG_COUNTER, rr_depth=5, max_depth=-1, hide_emitters=0,
                  block_size=32, block_first=0, block_stride=1, sample_first=0, sample_stride=1):
    """Defaults = the reference's effective integrator settings."""
    return RenderParams(spp, seed, rng_mode, rr_depth, max_depth, hide_emitters, block_size,
                        block_first, block_stride, sample_first, sample_stride)


class MskError(RuntimeError):
    """A non-zero status from the C ABI (the plugin side turns it into the reference's Throw)."""

    def __init__(self, code, text):
        super().__init__(f"msk_gpu error {code}: {text}")
        self.code = code


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSK_GPU_LIB") or os.path.join(_PKG_DIR, "lib", "libmsk_gpu.so")   # override: A/B experiments

# every symbol include/msk_gpu.h declares
EXPORTS = ["msk_gpu_init", "msk_gpu_shutdown", "msk_gpu_last_error", "msk_gpu_scene_create",
           "msk_gpu_scene_destroy", "msk_gpu_render", "msk_gpu_render_device", "msk_gpu_trace_closest",
           "msk_gpu_trace_any", "msk_gpu_sample_pixels", "msk_gpu_describe", "msk_gpu_render_aov", "msk_gpu_aov_channels"]

# integrators/aov.cpp:21-28
MSK_AOV_DEPTH, MSK_AOV_POSITION, MSK_AOV_UV, MSK_AOV_GEO_NORMAL, MSK_AOV_SH_NORMAL, MSK_AOV_PATH_RGBA = range(6)
AOV_WIDTH = (1, 3, 2, 3, 3, 4)

_lib = None


def load_library(path=None):
    """dlopen libmsk_gpu.so and type its entry points.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: the HIP back end is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'`).  There is no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, u64 = C.c_void_p, C.c_uint64
    lib.msk_gpu_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.msk_gpu_init.restype = C.c_int
    lib.msk_gpu_shutdown.argtypes = [vp]
    lib.msk_gpu_shutdown.restype = None
    lib.msk_gpu_last_error.argtypes = [vp]
    lib.msk_gpu_last_error.restype = C.c_char_p
    lib.msk_gpu_scene_create.argtypes = [vp, C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.msk_gpu_scene_create.restype = C.c_int
    lib.msk_gpu_scene_destroy.argtypes = [vp]
    lib.msk_gpu_scene_destroy.restype = None
    lib.msk_gpu_render.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.msk_gpu_render.restype = C.c_int
    lib.msk_gpu_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    lib.msk_gpu_render_device.restype = C.c_int
    lib.msk_gpu_trace_closest.argtypes = [vp, u64, vp, vp]
    lib.msk_gpu_trace_closest.restype = C.c_int
    lib.msk_gpu_trace_any.argtypes = [vp, u64, vp, vp]
    lib.msk_gpu_trace_any.restype = C.c_int
    lib.msk_gpu_sample_pixels.argtypes = [vp, C.POINTER(RenderParams), u64, vp, vp, vp]
    lib.msk_gpu_sample_pixels.restype = C.c_int
    lib.msk_gpu_render_aov.argtypes = [vp, C.POINTER(RenderParams), vp, C.c_uint32, vp, C.POINTER(Stats)]
    lib.msk_gpu_render_aov.restype = C.c_int
    lib.msk_gpu_aov_channels.argtypes = [vp, C.c_uint32]
    lib.msk_gpu_aov_channels.restype = C.c_uint32
    lib.msk_gpu_describe.argtypes = [vp, C.c_char_p, u64]
    lib.msk_gpu_describe.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """msk_ctx: one per process.  `device` is a HIP ordinal, or a sequence of ordinals for a GROUP context (msk_gpu_init with
    n > 1: renders are sample-sharded over the members and the films summed on the first device; an ordinal may repeat)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = C.c_void_p()
        try:
            devs = [operator.index(device)]               # any integer type (a numpy rank value included)
        except TypeError:
            devs = [operator.index(d) for d in device]
        self.devices = devs
        ids = (C.c_int * len(devs))(*devs)
        rc = self.lib.msk_gpu_init(ids, len(devs), C.byref(self.handle))
        if rc != MSK_OK:
            raise MskError(rc, (self.lib.msk_gpu_last_error(None) or b"").decode())

    def check(self, rc):
        if rc != MSK_OK:
            raise MskError(rc, (self.lib.msk_gpu_last_error(self.handle) or b"").decode())

    def describe(self):
        buf = C.create_string_buffer(1024)
        self.check(self.lib.msk_gpu_describe(self.handle, buf, 1024))
        return buf.value.decode()

    def close(self):
        if self.handle:
            self.lib.msk_gpu_shutdown(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Scene:
    """msk_scene: geometry + BVH + light tables resident in HBM."""

    def __init__(self, ctx, flat):
        """flat: hostmirror.FlatScene (keeps the numpy arrays the desc points into alive)."""
        self.ctx, self.flat = ctx, flat
        self.handle = C.c_void_p()
        ctx.check(ctx.lib.msk_gpu_scene_create(ctx.handle, C.byref(flat.desc), C.byref(self.handle)))

    @property
    def width(self):
        """of the film the render calls write: the crop window (the whole film by default)"""
        f = self.flat.desc.film
        return f.crop_size[0] if (f.crop_size[0] or f.crop_size[1]) else f.width

    @property
    def height(self):
        f = self.flat.desc.film
        return f.crop_size[1] if (f.crop_size[0] or f.crop_size[1]) else f.height

    def render(self, params, out=None):
        """-> (film float32[H,W,5] of weighted sums {X,Y,Z,A,W}, Stats).  out: an array to fill instead of a new one (a
        caller that renders repeatedly keeps its pages mapped: a fresh 5 MB array costs more in page faults than the copy)."""
        film = out if out is not None else np.empty((self.height, self.width, 5), np.float32)
        if not (isinstance(film, np.ndarray) and film.dtype == np.float32 and film.flags.c_contiguous and film.shape == (self.height, self.width, 5)):
            raise ValueError(f"out must be a C-contiguous float32 array of shape {(self.height, self.width, 5)}")      # the library writes that many bytes
        st = Stats()
        self.ctx.check(self.ctx.lib.msk_gpu_render(self.handle, C.byref(params), _ptr(film), C.byref(st)))
        return film, st

    def render_aov(self, params, aov_types):
        """The "aov" integrator: -> (film float32[H,W,5+C] of weighted sums {X,Y,Z,A,W, aov channels...}, Stats)."""
        types = np.ascontiguousarray(aov_types, np.int32)
        n_ch = self.ctx.lib.msk_gpu_aov_channels(_ptr(types), len(types)) if len(types) else 0
        film = np.empty((self.height, self.width, 5 + n_ch), np.float32)
        st = Stats()
        self.ctx.check(self.ctx.lib.msk_gpu_render_aov(self.handle, C.byref(params), _ptr(types), len(types), _ptr(film), C.byref(st)))
        return film, st

    def render_device(self, params, device_ptr, stream=None):
        st = Stats()
        self.ctx.check(self.ctx.lib.msk_gpu_render_device(self.handle, C.byref(params), C.c_void_p(device_ptr),
                                                           C.c_void_p(stream or 0), C.byref(st)))
        return st

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], 4), np.float32)
        self.ctx.check(self.ctx.lib.msk_gpu_trace_closest(self.handle, rays.shape[0], _ptr(rays), _ptr(out)))
        return out

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.empty(rays.shape[0], np.uint8)
        self.ctx.check(self.ctx.lib.msk_gpu_trace_any(self.handle, rays.shape[0], _ptr(rays), _ptr(out)))
        return out

    def sample_pixels(self, params, pixels):
        pixels = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
        n = pixels.shape[0]
        xyz = np.empty((n, params.spp, 3), np.float32)
        pos = np.empty((n, params.spp, 2), np.float32)
        self.ctx.check(self.ctx.lib.msk_gpu_sample_pixels(self.handle, C.byref(params), n, _ptr(pixels),
                                                           _ptr(xyz), _ptr(pos)))
        return xyz, pos

    def close(self):
        if self.handle:
            self.ctx.lib.msk_gpu_scene_destroy(self.handle)
            self.handle = C.c_void_p()
