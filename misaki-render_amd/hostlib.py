"""ctypes binding of the C++ host library (misaki-render_amd/lib/libmisaki-render.so): the reference's
scene-XML / Properties / plugin interface with the `"path"` integrator running on the MI355X."""
import ctypes as C
import os

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSK_HOST_LIB") or os.path.join(_PKG, "lib", "libmisaki-render.so")     # (override: a sanitizer build)
_lib = None


class HostError(RuntimeError):
    """A C++ exception (the reference's Throw) that reached the C boundary."""


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        abi.load_library()                       # libmsk_gpu.so first (RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.msk_host_last_error.restype = C.c_char_p
        lib.msk_host_load_scene.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
        lib.msk_host_free_scene.argtypes = [vp]
        lib.msk_host_flatten.argtypes = [vp, C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams)]
        lib.msk_host_render.argtypes = [vp, vp, vp, C.c_char_p, C.POINTER(abi.Stats)]
        lib.msk_host_film_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.msk_host_film_crop.argtypes = [vp, C.POINTER(C.c_int * 4)]
        lib.msk_host_aov_names.argtypes = [vp, C.c_char_p, C.c_size_t]
        lib.msk_host_aov_types.argtypes = [vp, vp, C.c_size_t]
        lib.msk_host_srgb_model_fetch.argtypes = [vp, vp]
        lib.msk_host_rgb2spec_build.argtypes = [C.c_int, C.c_char_p, C.c_int]
        lib.msk_host_srgb_model_source.argtypes = [C.c_char_p, C.c_size_t]
        lib.msk_host_write_image.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp]
        lib.msk_host_set_log_level.argtypes = [C.c_int]
        lib.msk_host_set_log_level(3)
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise HostError(load().msk_host_last_error().decode())


class HostFlat:
    """Same shape as hostmirror.FlatScene: .desc plus the numpy views tests use."""

    def __init__(self, desc, params, owner):
        self.desc, self.params, self.owner = desc, params, owner
        self.vertices = np.ctypeslib.as_array(desc.vertices, (desc.n_vertices, 8)) if desc.n_vertices else np.zeros((0, 8), np.float32)
        self.faces = np.ctypeslib.as_array(desc.faces, (desc.n_faces, 3)) if desc.n_faces else np.zeros((0, 3), np.uint32)


class HostScene:
    def __init__(self, xml_path, **params):
        self.lib = load()
        self.h = C.c_void_p()
        p = ";".join(f"{k}={v}" for k, v in params.items()).encode()
        _check(self.lib.msk_host_load_scene(str(xml_path).encode(), p, C.byref(self.h)))

    def film_size(self):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.msk_host_film_size(self.h, C.byref(w), C.byref(h), C.byref(s)))
        return w.value, h.value, s.value

    def film_crop(self):
        """(offset_x, offset_y, width, height) of the film's crop window = the storage a render fills"""
        a = (C.c_int * 4)()
        _check(self.lib.msk_host_film_crop(self.h, C.byref(a)))
        return tuple(a)

    def flatten(self):
        """The flatten step of the "path" plugin -> (msk_scene_desc, msk_render_params)."""
        d, p = abi.SceneDesc(), abi.RenderParams()
        _check(self.lib.msk_host_flatten(self.h, C.byref(d), C.byref(p)))
        return HostFlat(d, p, self)

    def aov_names(self):
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.msk_host_aov_names(self.h, buf, len(buf))
        if n < 0:
            _check(n)
        return [s for s in buf.value.decode().split("\n") if s]

    def aov_types(self):
        buf = np.zeros(64, np.int32)
        n = self.lib.msk_host_aov_types(self.h, buf.ctypes.data_as(C.c_void_p), len(buf))
        if n < 0:
            _check(n)
        return [int(x) for x in buf[:n]]

    def render(self, develop_to=None):
        """scene->integrator()->render(scene, sensor) on the GPU -> (film[H,W,5+C], image[H,W,4+C], Stats), H x W the film's crop window;
        C = the integrator's AOV channel count (0 for "path")."""
        _, _, w, h = self.film_crop()
        c = len(self.aov_names())
        film, rgba, st = np.zeros((h, w, 5 + c), np.float32), np.zeros((h, w, 4 + c), np.float32), abi.Stats()
        _check(self.lib.msk_host_render(self.h, film.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p),
                                        (develop_to or "").encode(), C.byref(st)))
        return film, rgba, st

    def close(self):
        if self.h:
            self.lib.msk_host_free_scene(self.h)
            self.h = C.c_void_p()


def srgb_model_fetch(rgb):
    a, o = np.asarray(rgb, np.float32), np.zeros(3, np.float32)
    _check(load().msk_host_srgb_model_fetch(a.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p)))
    return tuple(float(x) for x in o)


def rgb2spec_build(res, path, threads=0):
    """The res^3 x 3 sRGB upsampling table computed by the library's optimiser (host/src/rgb2spec_table.cpp) -> file."""
    _check(load().msk_host_rgb2spec_build(int(res), str(path).encode(), int(threads)))


def srgb_model_source():
    buf = C.create_string_buffer(4096)
    _check(load().msk_host_srgb_model_source(buf, len(buf)))
    return buf.value.decode()


def write_image(path, img):
    img = np.ascontiguousarray(img, np.float32)
    _check(load().msk_host_write_image(str(path).encode(), img.shape[1], img.shape[0], img.shape[2], img.ctypes.data_as(C.c_void_p)))
