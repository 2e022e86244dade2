"""Python mirror of the reference's HOST side of the hot path (setup only, no per-sample work).

What the reference's host does before SamplingIntegrator::render runs, restated so that
Python callers (tests, bench.py, smoke) can build the flattened `msk_scene_desc` the C ABI takes:

  * synthetic Cornell-box meshes (the reference ships none) and the
    OBJ loader's quad split + vertex layout (src/librender/shapes/obj.cpp:104-142)
  * PerspectiveCamera matrices (sensors/perspective.cpp:11-19, core/transform.h:169-187)
  * Gaussian reconstruction filter LUT (filters/gaussian.cpp:10-20, rfilter.cpp:12-27)
  * srgb / srgb_d65 texture parameters (spectra/srgb.cpp:13-19, srgb_d65.cpp:13-31) through
    this package's own Jakob-Hanika style spectral upsampling (rgb2spec.py)
  * HDRFilm::image() develop step (films/hdrfilm.cpp:48-90)

The C++ host library (misaki-render_amd/host/) provides the same through the reference's
plugin/Properties/XML interface; this module is the scripting-side equivalent.
"""
import ctypes as C
import json
import math
import os

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))


def cie_tables():
    """(cie1931_xyz float32[3*95], d65 float32[95]) — the tables the host hands to the back end."""
    d = json.load(open(os.path.join(_PKG, "data", "cie_tables.json")))
    xyz = np.array(d["cie1931_x"] + d["cie1931_y"] + d["cie1931_z"], np.float32)
    return xyz, np.array(d["d65"], np.float32)


# ----------------------------------------------------------------------------- filter
# ONE flattener truth (round 5): this module and the C++ host library (host/src/render.cpp, core.cpp) produce the same bits for
# the camera matrices and the filter table — tests/test_host_library.py holds them to np.array_equal at three film sizes.  What
# they share is the arithmetic, restated here operation for operation: the filter in fp32 with the C library's expf (numpy's
# own vectorised exp differs from it in the last place), the transforms in fp64 — Transform4f keeps a matrix and its inverse,
# products multiply both, inverse() swaps them, one rounding to fp32 at the end — which is NOT the reference's arithmetic
# to the bit: that is Eigen's fp32 4x4 product and its SSE cofactor inverse (transform.h:87-187), not reproducible without Eigen;
# the fp64 path is what those fp32 results approximate, and both of this repository's hosts now round it the same way.
_libm = C.CDLL("libm.so.6")
_libm.expf.restype, _libm.expf.argtypes = C.c_float, [C.c_float]


def _expf(x):
    return np.float32(_libm.expf(C.c_float(float(x))))


def gaussian_filter(stddev=0.5):
    """GaussianFilter ctor + ReconstructionFilter::init_discretization in fp32 (gaussian.cpp:10-20, rfilter.cpp:12-27), with the
    operations and the expf of host/src/render.cpp.  Returns (radius, lut[33])."""
    f = np.float32
    n = abi.MSK_FILTER_RESOLUTION
    stddev = f(stddev)
    radius = f(f(4) * stddev)
    alpha = f(f(-1.0) / f(f(f(2.0) * stddev) * stddev))
    bias = _expf(f(f(alpha * radius) * radius))
    vals = np.zeros(n + 1, np.float32)
    s = f(0)
    for i in range(n):
        x = f(f(radius * f(i)) / f(n))
        vals[i] = max(f(0), f(_expf(f(f(alpha * x) * x)) - bias))
        s = f(s + vals[i])
    s = f(s * f(f(f(2) * radius) / f(n)))
    norm = f(f(1.0) / s)
    for i in range(n):
        vals[i] = f(vals[i] * norm)
    return float(radius), vals


# ----------------------------------------------------------------------------- camera
class _T4:
    """Transform4f of the C++ host (core.h:64-80, core.cpp:80-175): a 4x4 matrix and its inverse in fp64."""

    def __init__(self, m, inv=None):
        self.m = [list(map(float, r)) for r in m]
        self.inv = [list(map(float, r)) for r in inv] if inv is not None else _T4._inverse(self.m)

    @staticmethod
    def _mul(a, b):                      # Matrix4f::operator*: sum over k = 0..3, left to right, from 0
        out = [[0.0] * 4 for _ in range(4)]
        for i in range(4):
            for j in range(4):
                acc = 0.0
                for k in range(4):
                    acc += a[i][k] * b[k][j]
                out[i][j] = acc
        return out

    @staticmethod
    def _inverse(m):                     # Matrix4f::inverse: Gauss-Jordan with partial pivoting
        a = [list(m[i]) + [1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
        for c in range(4):
            p = c
            for r in range(c + 1, 4):
                if abs(a[r][c]) > abs(a[p][c]):
                    p = r
            if a[p][c] == 0.0:
                return [[float("nan")] * 4 for _ in range(4)]
            if p != c:
                a[p], a[c] = a[c], a[p]
            inv = 1.0 / a[c][c]
            for j in range(8):
                a[c][j] *= inv
            for r in range(4):
                if r != c:
                    fct = a[r][c]
                    if fct != 0.0:
                        for j in range(8):
                            a[r][j] -= fct * a[c][j]
        return [a[i][4:] for i in range(4)]

    def __mul__(self, t):
        return _T4(_T4._mul(self.m, t.m), _T4._mul(t.inv, self.inv))

    def inverse(self):
        return _T4(self.inv, self.m)

    def to_float16(self, inverse=False):
        return np.array(self.inv if inverse else self.m, np.float64).astype(np.float32).reshape(16)

    @staticmethod
    def scale(v):
        v = [float(np.float32(x)) for x in v]
        return _T4([[v[0], 0, 0, 0], [0, v[1], 0, 0], [0, 0, v[2], 0], [0, 0, 0, 1]],
                   [[1.0 / v[0], 0, 0, 0], [0, 1.0 / v[1], 0, 0], [0, 0, 1.0 / v[2], 0], [0, 0, 0, 1]])

    @staticmethod
    def translate(v):
        v = [float(np.float32(x)) for x in v]
        return _T4([[1, 0, 0, v[0]], [0, 1, 0, v[1]], [0, 0, 1, v[2]], [0, 0, 0, 1]],
                   [[1, 0, 0, -v[0]], [0, 1, 0, -v[1]], [0, 0, 1, -v[2]], [0, 0, 0, 1]])

    @staticmethod
    def perspective(fov, near, far):
        fov, near, far = np.float32(fov), float(np.float32(near)), float(np.float32(far))
        recip = 1.0 / (far - near)
        cot = 1.0 / math.tan(float(fov / np.float32(2.0)) * (math.pi / 180.0))
        return _T4([[cot, 0, 0, 0], [0, cot, 0, 0], [0, 0, far * recip, -near * far * recip], [0, 0, 1, 0]])

    @staticmethod
    def lookat(origin, target, up):
        o, t, u = ([float(np.float32(x)) for x in v] for v in (origin, target, up))

        def nrm(v):
            l = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
            return [v[0] / l, v[1] / l, v[2] / l]

        def crs(a, b):
            return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
        d = nrm([t[0] - o[0], t[1] - o[1], t[2] - o[2]])
        left = nrm(crs(nrm(u), d))
        nup = nrm(crs(d, left))
        return _T4([[left[r], nup[r], d[r], o[r]] for r in range(3)] + [[0, 0, 0, 1]])


def perspective_camera(fov, near, far, width, height, origin, target, up):
    """-> (sample_to_camera[16], to_world[16]) row-major float32; sample space is in pixels.

    perspective.cpp:11-19: camera_to_sample = S(w,h,1) S(-.5,-.5a,1) T(-1,-1/a,0) P(fov,near,far), sample_to_camera its inverse;
    to_world = lookat (transform.h:169-178).  The C++ host's arithmetic (see the note above gaussian_filter), bit for bit.
    """
    f = np.float32
    aspect = f(f(width) / f(height))                 # sensor.cpp: size.x / (float) size.y
    c2s = _T4.scale((f(width), f(height), f(1))) * _T4.scale((f(-0.5), f(f(-0.5) * aspect), f(1))) * \
        _T4.translate((f(-1), f(f(-1) / aspect), f(0))) * _T4.perspective(fov, near, far)
    return c2s.inverse().to_float16(), _T4.lookat(origin, target, up).to_float16()


# ----------------------------------------------------------------------------- meshes
class MeshSpec:
    """One <shape type="obj"> of the scene: faces as 3- or 4-tuples of 3D points."""

    def __init__(self, name, faces, reflectance, radiance=None, translate=(0, 0, 0), normals=None, bsdf=None, texcoords=None):
        """bsdf: None = <bsdf type="diffuse"> with `reflectance`; or a dict for a rough conductor
        {"type": "roughconductor", "alpha": a | ("alpha_u","alpha_v"), "eta": rgb, "k": rgb,
         "specular_reflectance": rgb (default 1), "sample_visible": bool, "twosided": bool}, a rough dielectric, or
        {"type": "diffuse", "twosided": bool, "texture": {"type": "checkerboard", "color0": rgb, "color1": rgb,
         "scale": (sx, sy) | "matrix": 16 floats (the to_uv 4x4, row-major)}} (texture absent = `reflectance`).
        texcoords: None, or per face a tuple of (u, v) per corner as written in the OBJ `vt` lines (the loader stores 1 - v)."""
        self.name, self.faces, self.reflectance, self.radiance = name, faces, reflectance, radiance
        self.translate, self.normals, self.bsdf, self.texcoords = translate, normals, bsdf, texcoords


def triangulate(mesh):
    """OBJ loader semantics: 8 floats per vertex, quad (v0,v1,v2,v3) -> (v0,v1,v2),(v3,v0,v2)
    (obj.cpp:104-133); `to_world` translate applied to positions at load (obj.cpp:90)."""
    verts, faces = [], []
    tr = np.asarray(mesh.translate, np.float32)
    for fi, f in enumerate(mesh.faces):
        base = len(verts)
        for ci, p in enumerate(f):
            # Transform4f::apply_point with a pure translation: p + t in fp32
            q = np.asarray(p, np.float32) + tr
            uv = mesh.texcoords[fi][ci] if mesh.texcoords is not None else (0, 0)
            verts.append([q[0], q[1], q[2], 0, 0, 0, np.float32(uv[0]),
                          (np.float32(1) - np.float32(uv[1])) if mesh.texcoords is not None else 0])   # flip_tex_coords, obj.cpp:96-97
        if len(f) == 3:
            faces.append([base, base + 1, base + 2])
        else:
            faces.append([base, base + 1, base + 2])
            faces.append([base + 3, base, base + 2])
    return np.array(verts, np.float32).reshape(-1, 8), np.array(faces, np.uint32).reshape(-1, 3)


def write_obj(mesh, path):
    """The same geometry as an OBJ file the reference's loader (and the C++ host) can read."""
    with open(path, "w") as fh:
        fh.write(f"# {mesh.name}: synthetic Cornell-box mesh (classic Cornell data)\n")
        n = 0
        for fi, f in enumerate(mesh.faces):
            for p in f:
                fh.write("v %.9g %.9g %.9g\n" % tuple(float(np.float32(c)) for c in p))
            if mesh.texcoords is not None:
                for uv in mesh.texcoords[fi]:
                    fh.write("vt %.9g %.9g\n" % tuple(float(np.float32(c)) for c in uv))
                fh.write("f " + " ".join("%d/%d" % (n + i + 1, n + i + 1) for i in range(len(f))) + "\n")
            else:
                fh.write("f " + " ".join(str(n + i + 1) for i in range(len(f))) + "\n")
            n += len(f)


def _bsdf_xml(m, v3):
    """The <bsdf> element of one shape (reference plugin parameter names)."""
    spec = m.bsdf
    refl_xml = '<spectrum name="reflectance" value="%s"/>' % m.reflectance.text if isinstance(m.reflectance, Regular) else \
               '<spectrum name="reflectance" value="%.9g"/>' % float(m.reflectance) if np.isscalar(m.reflectance) else \
               '<rgb name="reflectance" value="%s"/>' % v3(m.reflectance)
    if spec is None:
        return ['        <bsdf type="diffuse">', '            ' + refl_xml, '        </bsdf>']
    if spec["type"] == "diffuse":
        tex = spec.get("texture")
        if tex is None:
            body = [refl_xml]
        else:     # textures/checkerboard.cpp:11-15 parameter names, as in results/Figure_2_RoughConductor/roughconductor.xml:35-41
            xf = '<scale x="%.9g" y="%.9g"/>' % tuple(tex["scale"]) if "scale" in tex else \
                 '<matrix value="%s"/>' % " ".join("%.9g" % float(np.float32(x)) for x in tex["matrix"])
            body = ['<texture name="reflectance" type="%s">' % tex["type"], '    <rgb name="color0" value="%s"/>' % v3(tex["color0"]),
                    '    <rgb name="color1" value="%s"/>' % v3(tex["color1"]), '    <transform name="to_uv">', '        ' + xf,
                    '    </transform>', '</texture>']
        inner = ['<bsdf type="diffuse">'] + ['    ' + b for b in body] + ['</bsdf>']
        if spec.get("twosided"):
            inner = ['<bsdf type="twosided">'] + ['    ' + b for b in inner] + ['</bsdf>']
        return ['        ' + b for b in inner]
    body = []
    a = spec.get("alpha", 0.1)
    if np.isscalar(a):
        body.append('<float name="alpha" value="%r"/>' % float(a))
    else:
        body += ['<float name="alpha_u" value="%r"/>' % float(a[0]), '<float name="alpha_v" value="%r"/>' % float(a[1])]
    body.append('<string name="distribution" value="ggx"/>')
    if spec.get("sample_visible"):
        body.append('<boolean name="sample_visible" value="true"/>')
    keys = {"roughconductor": ("eta", "k", "specular_reflectance"),
            "roughdielectric": ("specular_reflectance", "specular_transmittance")}[spec["type"]]
    for k in keys:
        if k in spec:
            body.append('<spectrum name="%s" value="%s"/>' % (k, spec[k].text) if isinstance(spec[k], Regular) else
                        '<rgb name="%s" value="%s"/>' % (k, v3(spec[k])))
    for k in ("int_ior", "ext_ior"):
        if k in spec:
            body.append('<float name="%s" value="%r"/>' % (k, float(spec[k])))
    inner = ['<bsdf type="%s">' % spec["type"]] + ['    ' + b for b in body] + ['</bsdf>']
    if spec.get("twosided"):
        inner = ['<bsdf type="twosided">'] + ['    ' + b for b in inner] + ['</bsdf>']
    return ['        ' + b for b in inner]


def write_scene_xml(meshes, directory, width, height, spp, camera=None, integrator_props=None, film_type="hdrfilm",
                    filename="scene.xml", env=None, film_props=None):
    """Writes <directory>/meshes/*.obj and a Mitsuba-style scene XML the C++ host (and the reference's
    loader) understands; same structure as results/Figure_1_Pathtrace/scene.xml, $-parameters for spp/size."""
    camera = camera or CBOX_CAMERA
    os.makedirs(os.path.join(directory, "meshes"), exist_ok=True)
    v3 = lambda v: ", ".join("%.9g" % float(x) for x in v)
    out = ['<scene>', '    <default name="spp" value="%d"/>' % spp, '    <default name="width" value="%d"/>' % width,
           '    <default name="height" value="%d"/>' % height, '    <integrator type="path">']
    for k, val in (integrator_props or {}).items():
        tag = "boolean" if isinstance(val, bool) else "string" if isinstance(val, str) else "integer"
        out.append('        <%s name="%s" value="%s"/>' % (tag, k, str(val).lower()))
    out += ['    </integrator>', '    <sensor type="perspective">',
            '        <float name="near_clip" value="%.9g"/>' % camera["near"], '        <float name="far_clip" value="%.9g"/>' % camera["far"],
            '        <float name="fov" value="%.9g"/>' % camera["fov"], '        <transform name="to_world">',
            '            <lookat origin="%s" target="%s" up="%s"/>' % (v3(camera["origin"]), v3(camera["target"]), v3(camera["up"])),
            '        </transform>', '        <sampler type="independent">', '            <integer name="sample_count" value="$spp"/>',
            '        </sampler>', '        <film type="%s">' % film_type, '            <integer name="width" value="$width"/>',
            '            <integer name="height" value="$height"/>'] + \
           ['            <integer name="%s" value="%d"/>' % (k, int(v)) for k, v in (film_props or {}).items()] + ['        </film>', '    </sensor>']

    def env_xml():
        rad = env.get("radiance")
        body = ['        <spectrum name="radiance" value="%s"/>' % rad.text] if isinstance(rad, Regular) else \
               ['        <rgb name="radiance" value="%s"/>' % v3(rad)] if rad is not None else \
               (['        <spectrum name="radiance" value="%r"/>' % float(env["scale"])] if "scale" in env else [])   # D65 * scale
        return ['    <emitter type="constant">'] + body + ['    </emitter>']
    if env is not None and env.get("first"):
        out += env_xml()
    for m in meshes:
        write_obj(m, os.path.join(directory, "meshes", m.name + ".obj"))
        out += ['    <shape type="obj">', '        <string name="filename" value="meshes/%s.obj"/>' % m.name]
        if any(float(t) != 0 for t in m.translate):
            out += ['        <transform name="to_world">', '            <translate x="%.9g" y="%.9g" z="%.9g"/>' % tuple(m.translate),
                    '        </transform>']
        out += _bsdf_xml(m, v3)
        if m.radiance is not None:
            out += ['        <emitter type="area">', '            <spectrum name="radiance" value="%s"/>' % m.radiance.text if isinstance(m.radiance, Regular)
                    else '            <rgb name="radiance" value="%s"/>' % v3(m.radiance), '        </emitter>']
        out.append('    </shape>')
    if env is not None and not env.get("first"):
        out += env_xml()
    out.append('</scene>')
    path = os.path.join(directory, filename)
    open(path, "w").write("\n".join(out) + "\n")
    return path


WHITE = (0.885809, 0.698859, 0.666422)
GREEN = (0.105421, 0.37798, 0.076425)
RED = (0.570068, 0.0430135, 0.0443706)
BOX = (0.45, 0.30, 0.90)
LUMINAIRE = (0.936461, 0.740433, 0.705267)


def cbox_meshes():
    """The 8 shapes of results/Figure_1_Pathtrace/scene.xml:33-103 in XML order (= geomID order),
    geometry from the classic Cornell measurements."""
    q = lambda *p: tuple(p)
    return [
        MeshSpec("cbox_luminaire", [q((343, 548.8, 227), (343, 548.8, 332), (213, 548.8, 332), (213, 548.8, 227))],
                 LUMINAIRE, radiance=(40, 40, 40), translate=(0, -0.5, 0)),
        MeshSpec("cbox_floor", [q((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2))], WHITE),
        MeshSpec("cbox_ceiling", [q((556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0))], WHITE),
        MeshSpec("cbox_back", [q((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2))], WHITE),
        MeshSpec("cbox_greenwall", [q((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2))], GREEN),
        MeshSpec("cbox_redwall", [q((552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0))], RED),
        MeshSpec("cbox_smallbox", [
            q((130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)),
            q((290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)),
            q((130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)),
            q((82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)),
            q((240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225))], BOX),
        MeshSpec("cbox_largebox", [
            q((423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)),
            q((423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)),
            q((472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)),
            q((314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)),
            q((265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247))], BOX),
    ]


CBOX_CAMERA = dict(fov=49.3077, near=10.0, far=2800.0, origin=(278, 273, -800), target=(278, 273, -799),
                   up=(0, 1, 0))


def blob_mesh(name, center, radius, n_theta, n_phi, reflectance, seed=1, bump=0.15):
    """A closed, bumpy, outward-wound triangle mesh (bunny/teapot-class stand-in):
    a UV sphere displaced by a few low-frequency sinusoids.  ~2*n_theta*n_phi triangles."""
    rng = np.random.RandomState(seed)
    k = rng.uniform(1, 4, (4, 2)).round()
    ph = rng.uniform(0, 2 * np.pi, 4)

    def point(i, j):
        th = np.pi * i / n_theta
        p = 2 * np.pi * (j % n_phi) / n_phi
        r = 1.0
        if 0 < i < n_theta:
            for a in range(4):
                r += bump / 4 * np.sin(k[a, 0] * th * 2 + ph[a]) * np.cos(k[a, 1] * p + ph[a])
        d = np.array([np.sin(th) * np.cos(p), np.cos(th), np.sin(th) * np.sin(p)])
        return tuple(np.asarray(center, np.float64) + radius * r * d)

    faces = []
    for i in range(n_theta):
        for j in range(n_phi):
            a, b, c, d = point(i, j), point(i + 1, j), point(i + 1, j + 1), point(i, j + 1)
            if i == 0:
                faces.append((a, c, b))
            elif i == n_theta - 1:
                faces.append((a, d, b))
            else:
                faces.append((a, d, c))
                faces.append((a, c, b))
    return MeshSpec(name, faces, reflectance)


MSK_CIE_Y_NORMALIZATION = np.float32(1.0 / 106.7502593994140625)       # include/misaki/core/spectrum.h:75


class Regular:
    """A tabulated spectrum on a regular grid = the reference's `regular` texture (spectra/regular.cpp:27-91): what
    <spectrum value="l0:v0, l1:v1, ..."/> with equidistant wavelengths makes (xml.cpp:300-341).  Accepted wherever the mirror takes
    an rgb: MeshSpec.reflectance, MeshSpec.radiance, a BSDF's eta / k / specular_*, the environment's radiance."""

    def __init__(self, lambda_min, lambda_max, values, text=None):
        self.lambda_min, self.lambda_max = float(np.float32(lambda_min)), float(np.float32(lambda_max))
        self.values = np.asarray(values, np.float32).copy()
        self.text = text          # the "l:v, l:v, ..." it was parsed from (write_scene_xml writes it back)

    @staticmethod
    def from_pairs(text, within_emitter=False):
        """create_texture_from_spectrum (xml.cpp:279-341) for "l:v, l:v, ...": values x MSK_CIE_Y_NORMALIZATION inside an
        <emitter>; the steps must agree within math::Epsilon (else the reference makes an `irregular` spectrum, which the back
        end does not take)."""
        pairs = [t.split(":") for t in text.replace(",", " ").split()]
        wl = np.array([np.float32(a) for a, _ in pairs], np.float32)
        v = np.array([np.float32(b) for _, b in pairs], np.float32)
        if within_emitter:
            v = v * MSK_CIE_Y_NORMALIZATION
        step = np.diff(wl)
        if (step < 0).any():
            raise ValueError("Wavelengths must be specified in increasing order!")
        if len(wl) > 2 and (np.abs(step[1:] - step[0]) > np.float32(2.0 ** -24)).any():
            raise ValueError("irregular spectrum (unequal wavelength steps): not supported by the GPU path integrator")
        return Regular(wl[0], wl[-1], v, text=text)


class _RegularPool:
    """The scene's regular_spectra / regular_values arrays while a scene is flattened."""

    def __init__(self):
        self.descs, self.values = [], []

    def add(self, r):
        n = len(r.values)
        if not (2 <= n <= abi.MSK_REGULAR_MAX):
            raise ValueError("a regular spectrum needs 2..%d values (got %d)" % (abi.MSK_REGULAR_MAX, n))
        self.descs.append(abi.RegularSpectrumDesc(r.lambda_min, r.lambda_max, n, len(self.values)))
        self.values += [float(x) for x in r.values]
        return len(self.descs)


def spectrum_desc(rgb, fetch, pool=None):
    """rgb -> msk_spectrum_desc.  In-gamut colours: S(fetch(rgb)); values above 1 (conductor eta/k) use the
    normalisation of spectra/srgb_d65.cpp:18-22 without the D65 factor: scale = 2 max, fetch(rgb / scale).
    A Regular: the tabulated form (its index in the scene's pool)."""
    if isinstance(rgb, Regular):
        return abi.SpectrumDesc((C.c_float * 3)(0.0, 0.0, 0.0), 1.0, pool.add(rgb))
    rgb = np.asarray(rgb, np.float32)
    if rgb.max() <= 1.0:
        return abi.SpectrumDesc((C.c_float * 3)(*fetch(tuple(float(x) for x in rgb))), 1.0)
    scale = np.float32(rgb.max() * np.float32(2.0))
    return abi.SpectrumDesc((C.c_float * 3)(*fetch(tuple(float(x) for x in rgb / scale))), float(scale))


def _texture_desc(tex, fetch):
    """textures/checkerboard.cpp:11-15: m_transform = the top-left 3x3 of the to_uv 4x4 (transform.h:142-148)."""
    if tex["type"] != "checkerboard":
        raise ValueError(tex["type"])
    t = abi.TextureDesc()
    t.type = abi.MSK_TEXTURE_CHECKERBOARD
    t.color0[:] = fetch(tuple(tex["color0"]))
    t.color1[:] = fetch(tuple(tex["color1"]))
    if "scale" in tex:
        m4 = np.diag([tex["scale"][0], tex["scale"][1], 1.0, 1.0]).astype(np.float32)
    else:
        m4 = np.asarray(tex["matrix"], np.float32).reshape(4, 4)
    t.to_uv[:] = [float(x) for x in (m4[0, 0], m4[0, 1], m4[0, 2], m4[1, 0], m4[1, 1], m4[1, 2])]
    return t


def _bsdf_desc(m, fetch, index, textures=None, pool=None):
    b = abi.BsdfDesc()
    b.back_bsdf = -1
    one = abi.SpectrumDesc((C.c_float * 3)(0.0, 0.0, float("inf")), 1.0)
    b.eta, b.k, b.specular_reflectance, b.specular_transmittance = one, one, one, one
    b.ior_eta = b.ior_inv_eta = 1.0
    b.reflectance_scale = 1.0
    spec = m.bsdf or {"type": "diffuse"}
    if spec["type"] == "diffuse":
        b.type = abi.MSK_BSDF_DIFFUSE
        if isinstance(m.reflectance, Regular):
            b.reflectance_regular = pool.add(m.reflectance)
        elif np.isscalar(m.reflectance):        # <spectrum name="reflectance" value="c"/>: the `uniform` plugin (spectra/uniform.cpp)
            b.reflectance[:] = (0.0, 0.0, float("inf"))
            b.reflectance_scale = float(m.reflectance)
        else:
            b.reflectance[:] = fetch(tuple(m.reflectance))
            b.reflectance_scale = 1.0
        if spec.get("texture") is not None:
            textures.append(_texture_desc(spec["texture"], fetch))
            b.reflectance_texture = len(textures)
    elif spec["type"] == "roughconductor":
        b.type = abi.MSK_BSDF_ROUGHCONDUCTOR
        a = spec.get("alpha", 0.1)
        b.alpha_u, b.alpha_v = (a, a) if np.isscalar(a) else a
        b.sample_visible = int(bool(spec.get("sample_visible", False)))
        b.eta, b.k = spectrum_desc(spec["eta"], fetch, pool), spectrum_desc(spec["k"], fetch, pool)
        b.specular_reflectance = spectrum_desc(spec.get("specular_reflectance", (1.0, 1.0, 1.0)), fetch, pool)
    elif spec["type"] == "roughdielectric":
        # bsdfs/roughdielectric.cpp:14-24: m_eta = int_ior / ext_ior in fp32
        b.type = abi.MSK_BSDF_ROUGHDIELECTRIC
        a = spec.get("alpha", 0.1)
        b.alpha_u, b.alpha_v = (a, a) if np.isscalar(a) else a
        b.sample_visible = int(bool(spec.get("sample_visible", False)))
        int_ior, ext_ior = np.float32(spec.get("int_ior", 1.5046)), np.float32(spec.get("ext_ior", 1.00028))
        b.ior_eta, b.ior_inv_eta = float(int_ior / ext_ior), float(ext_ior / int_ior)
        b.specular_reflectance = spectrum_desc(spec.get("specular_reflectance", (1.0, 1.0, 1.0)), fetch, pool)
        b.specular_transmittance = spectrum_desc(spec.get("specular_transmittance", (1.0, 1.0, 1.0)), fetch, pool)
    else:
        raise ValueError(spec["type"])
    if spec.get("twosided"):
        b.back_bsdf = index            # twosided(A): the same BSDF on both sides (twosided.cpp:23-24)
    return b


# ----------------------------------------------------------------------------- flatten
class FlatScene:
    """Owns the numpy arrays an msk_scene_desc points into."""

    def __init__(self):
        self.desc = abi.SceneDesc()
        self.keep = []


def _radiance_desc(radiance, fetch, scale_in=1.0):
    """srgb_d65 (spectra/srgb_d65.cpp:13-31): scale = 2*max(rgb); color /= scale; d65 scale *= scale;
    d65.cpp:33-34: m_scale *= 1/10568.  radiance None = Texture::D65(scale_in) (constant.cpp:16)."""
    if radiance is None:
        return (0.0, 0.0, float("inf")), float(np.float32(scale_in) * (np.float32(1.0) / np.float32(10568.0)))
    rad = np.asarray(radiance, np.float32)
    scale = np.float32(rad.max() * np.float32(2.0))
    col = rad / scale if scale != 0 else rad
    ce = fetch(tuple(float(x) for x in col))
    return ce, float(np.float32(np.float32(scale_in) * scale) * (np.float32(1.0) / np.float32(10568.0)))


def flatten(meshes, width, height, camera=None, filter_stddev=0.5, coeff_lookup=None, env=None, crop=None):
    """Scene -> msk_scene_desc, the step the `"path"` plugin's render() performs before calling
    the C ABI.  coeff_lookup(rgb)->(c0,c1,c2) overrides the spectral upsampling
    (tests pass the reference's own rgb2spec_fetch results); default = this package's rgb2spec.
    env: None, or {"radiance": rgb | None (= D65), "scale": 1.0, "first": False} = a top-level
    <emitter type="constant"> placed after (or, with first=True, before) the shapes in the XML.
    crop: None, or (offset_x, offset_y, width, height) = the film's crop_offset_x/_y, crop_width/_height (film.cpp:12-21)."""
    from . import rgb2spec
    fetch = coeff_lookup or rgb2spec.srgb_model_fetch
    camera = camera or CBOX_CAMERA
    fs = FlatScene()
    all_v, all_f, md, bd, ed, td = [], [], [], [], [], []
    pool = _RegularPool()
    nv = nf = 0

    def emitter_desc(kind, mesh_id, radiance, scale_in=1.0):
        if isinstance(radiance, Regular):        # a `regular` radiance as it stands (no D65 factor, no sigmoid)
            return abi.EmitterDesc(kind, mesh_id, (C.c_float * 3)(0.0, 0.0, 0.0), 0.0, pool.add(radiance))
        ce, sc = _radiance_desc(radiance, fetch, scale_in)
        return abi.EmitterDesc(kind, mesh_id, (C.c_float * 3)(*ce), float(sc), 0)

    def env_desc():
        return emitter_desc(abi.MSK_EMITTER_CONSTANT, -1, env.get("radiance"), env.get("scale", 1.0))
    if env is not None and env.get("first"):
        ed.append(env_desc())
    for i, m in enumerate(meshes):
        v, f = triangulate(m)
        bd.append(_bsdf_desc(m, fetch, len(bd), td, pool))
        eid = -1
        if m.radiance is not None:
            ed.append(emitter_desc(abi.MSK_EMITTER_AREA, i, m.radiance))
            eid = len(ed) - 1
        md.append(abi.MeshDesc(nv, len(v), nf, len(f), i, eid, 0, 1 if m.texcoords is not None else 0))
        all_v.append(v)
        all_f.append(f)
        nv += len(v)
        nf += len(f)
    if env is not None and not env.get("first"):
        ed.append(env_desc())
    verts = np.ascontiguousarray(np.concatenate(all_v + [np.zeros((0, 8), np.float32)]), np.float32).reshape(-1, 8)
    faces = np.ascontiguousarray(np.concatenate(all_f + [np.zeros((0, 3), np.uint32)]), np.uint32).reshape(-1, 3)
    cie, d65 = cie_tables()
    meshes_a = (abi.MeshDesc * max(1, len(md)))(*md)
    bsdfs_a = (abi.BsdfDesc * max(1, len(bd)))(*bd)
    emit_a = (abi.EmitterDesc * max(1, len(ed)))(*ed)
    tex_a = (abi.TextureDesc * max(1, len(td)))(*td)
    reg_a = (abi.RegularSpectrumDesc * max(1, len(pool.descs)))(*pool.descs)
    reg_v = np.ascontiguousarray(np.array(pool.values + ([] if pool.values else [0.0]), np.float32))
    fs.keep += [verts, faces, cie, d65, meshes_a, bsdfs_a, emit_a, tex_a, reg_a, reg_v]
    d = fs.desc
    d.abi_version = abi.MSK_ABI_VERSION
    d.n_meshes, d.n_bsdfs, d.n_emitters = len(md), len(bd), len(ed)
    d.meshes, d.bsdfs, d.emitters = meshes_a, bsdfs_a, emit_a
    d.n_textures, d.textures = len(td), tex_a
    d.n_regular_spectra, d.n_regular_values = len(pool.descs), len(pool.values)
    d.regular_spectra = reg_a
    d.regular_values = reg_v.ctypes.data_as(C.POINTER(C.c_float))
    d.vertices = verts.ctypes.data_as(C.POINTER(C.c_float))
    d.faces = faces.ctypes.data_as(C.POINTER(C.c_uint32))
    d.n_vertices, d.n_faces = nv, nf
    s2c, tw = perspective_camera(camera["fov"], camera["near"], camera["far"], width, height,
                                 camera["origin"], camera["target"], camera["up"])
    d.camera.sample_to_camera[:] = s2c.tolist()
    d.camera.to_world[:] = tw.tolist()
    d.camera.near_clip, d.camera.far_clip = camera["near"], camera["far"]
    radius, lut = gaussian_filter(filter_stddev)
    d.film.width, d.film.height, d.film.filter_radius = width, height, radius
    d.film.filter_lut[:] = lut.tolist()
    if crop is not None:
        d.film.crop_offset[:] = [int(crop[0]), int(crop[1])]
        d.film.crop_size[:] = [int(crop[2]), int(crop[3])]
    d.cie1931_xyz = cie.ctypes.data_as(C.POINTER(C.c_float))
    d.d65 = d65.ctypes.data_as(C.POINTER(C.c_float))
    fs.vertices, fs.faces = verts, faces
    return fs


def cbox_scene(width, height, coeff_lookup=None, extra_meshes=(), crop=None):
    return flatten(cbox_meshes() + list(extra_meshes), width, height, coeff_lookup=coeff_lookup, crop=crop)


# BASELINE configs 3 and 5 name assets/bunny and assets/teapot-full, whose meshes the reference does not ship
# (SURVEY F3): the stand-ins are closed bumpy meshes of the same triangle class inside the Cornell room.
def bunny_class_scene(size, res=187, crop=None):
    """Config-3 class: ~70 k-triangle rough-conductor mesh (two-sided), room + luminaire, no boxes."""
    blob = blob_mesh("bunny_class", (278, 200, 280), 160, res, res, WHITE, seed=7, bump=0.25)
    blob.bsdf = {"type": "roughconductor", "alpha": 0.15, "eta": (0.2, 0.92, 1.1), "k": (3.9, 2.45, 2.14), "twosided": True}
    return flatten(cbox_meshes()[:6] + [blob], size, size, crop=crop)


def teapot_class_scene(size, res=270, diffuse=False, crop=None):
    """Config-5 class: ~146 k-triangle rough-dielectric mesh, room + luminaire, no boxes (diffuse=True: the same geometry,
    white diffuse — measurements of the shading variants on one scene)."""
    blob = blob_mesh("teapot_class", (278, 200, 280), 160, res, res, WHITE, seed=7, bump=0.25)
    if not diffuse:
        blob.bsdf = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
    return flatten(cbox_meshes()[:6] + [blob], size, size, crop=crop)


# ----------------------------------------------------------------------------- develop
_XYZ_TO_SRGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556],
                         [0.055648, -0.204043, 1.057311]], np.float32)


def develop(film_xyzaw):
    """HDRFilm::image() (hdrfilm.cpp:48-90): rgb = xyz_to_srgb(XYZ) / W, alpha = A / W -> float32[H,W,4]."""
    f = np.asarray(film_xyzaw, np.float32)
    xyz = f[..., :3]
    # Eigen 3x3 * vec3 row reduction, a0 + (a1 + a2)
    rgb = np.stack([_XYZ_TO_SRGB[r, 0] * xyz[..., 0] + (_XYZ_TO_SRGB[r, 1] * xyz[..., 1] +
                                                        _XYZ_TO_SRGB[r, 2] * xyz[..., 2]) for r in range(3)], -1)
    w = f[..., 4]
    inv = np.where(w != 0, np.float32(1.0) / np.where(w != 0, w, 1), np.float32(0)).astype(np.float32)
    return np.concatenate([rgb * inv[..., None], (f[..., 3] * inv)[..., None]], -1).astype(np.float32)
