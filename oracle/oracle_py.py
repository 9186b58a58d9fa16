"""TEST INFRASTRUCTURE ONLY: ctypes front-end of oracle/liboracle.so (see oracle.cpp's header).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
reported baseline.  Never imported by pathtracer_amd/.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PIXEL_DTYPE = np.dtype([('color', '<f4', 3), ('n', '<f4'), ('sum', '<f4', 3)])   # rendering.d:24-30
assert PIXEL_DTYPE.itemsize == 28


class OrcConfig(C.Structure):
    _fields_ = [('width', C.c_uint32), ('height', C.c_uint32), ('mode', C.c_uint32), ('hasSky', C.c_uint32),
                ('hasLights', C.c_uint32), ('sunDir', C.c_float * 3), ('sunRadiance', C.c_float * 3),
                ('skyRadiance', C.c_float * 3), ('floatEpsilon', C.c_float), ('rrProb', C.c_float),
                ('maxDepth', C.c_uint32), ('numIndirect', C.c_uint32), ('seed', C.c_uint32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle/liboracle.so is not built: run `make -C oracle`')
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_config.argtypes = [C.c_void_p, C.POINTER(OrcConfig)]
        L.orc_set_camera.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_float, C.c_float]
        L.orc_camera_lookat.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tree_build_seconds.restype = C.c_double
        L.orc_tree_build_seconds.argtypes = [C.c_void_p]
        L.orc_num_nodes.restype = C.c_uint32; L.orc_num_nodes.argtypes = [C.c_void_p]
        L.orc_num_light_groups.restype = C.c_uint32; L.orc_num_light_groups.argtypes = [C.c_void_p]
        L.orc_light_group.restype = C.c_uint32
        L.orc_light_group.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_check_tree.restype = C.c_int; L.orc_check_tree.argtypes = [C.c_void_p]
        L.orc_trace_primary.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 4
        L.orc_frontface_disagreements.restype = C.c_uint32
        L.orc_frontface_disagreements.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_view_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.orc_interp_normal.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_set_texcoords.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_trace_tex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int] + [C.c_void_p] * 4
        L.orc_render_mt.restype = C.c_double
        L.orc_render_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_render_keyed.restype = C.c_double
        L.orc_render_keyed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_int, C.c_void_p]
        L.orc_sincos.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_set_libm_trig.argtypes = [C.c_int]
        L.orc_pow_inv24.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_angle_directions.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_keyed_stream.argtypes = [C.c_uint32] * 5 + [C.c_void_p]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_bvh_count.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def camera_lookat(fov_deg, aspect, frm, to, up):
    """camera.d:16-29 -> float32[14] = pos dir right up tanFov aspect."""
    out = np.zeros(14, np.float32)
    f, t, u = (np.asarray(x, np.float32) for x in (frm, to, up))
    lib().orc_camera_lookat(fov_deg, aspect, _p(f), _p(t), _p(u), _p(out))
    return out


class Oracle:
    """Scene + integrator of the restated reference algorithm."""

    def __init__(self, flat, build_tree=True):
        self.v = np.ascontiguousarray(flat['v'], np.float32).reshape(-1, 9)
        self.nrm = np.ascontiguousarray(flat['n'], np.float32).reshape(-1, 9)
        self.mi = np.ascontiguousarray(flat['mat_idx'], np.uint32)
        self.mats = np.ascontiguousarray(flat['materials'], np.float32).reshape(-1, 4)
        self.T, self.M = len(self.v), len(self.mats)
        self.h = lib().orc_create(_p(self.v), _p(self.nrm), _p(self.mi), _p(self.mats), self.T, self.M, int(build_tree))
        self.cfg = OrcConfig()
        self.has_tree = bool(build_tree)
        if flat.get('tex') is not None:
            self.tex = np.ascontiguousarray(flat['tex'], np.float32).reshape(-1, 6)
            lib().orc_set_texcoords(self.h, _p(self.tex))
        self.configure()

    def close(self):
        if self.h:
            lib().orc_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, width=320, height=240, mode=0, has_sky=False, has_lights=True, eps=1e-5, rr=0.7, max_depth=5,
                  n_indirect=5, seed=0, sun_dir=None, sun_radiance=(1.0, 0.984, 0.8), sky_radiance=None):
        c = self.cfg
        c.width, c.height, c.mode, c.hasSky, c.hasLights = width, height, mode, int(has_sky), int(has_lights)
        if sun_dir is None:                                   # rendering.d:72  vec3(1,1,3).normalized()
            sun_dir = default_sun_dir()
        if sky_radiance is None:                              # rendering.d:74  vec3(.682,.977,1)*0.1
            sky_radiance = default_sky_radiance()
        c.sunDir[:] = [float(x) for x in sun_dir]; c.sunRadiance[:] = [float(x) for x in sun_radiance]
        c.skyRadiance[:] = [float(x) for x in sky_radiance]
        c.floatEpsilon, c.rrProb, c.maxDepth, c.numIndirect, c.seed = eps, rr, max_depth, n_indirect, seed
        lib().orc_set_config(self.h, C.byref(c))

    def set_camera(self, cam14):
        cam14 = np.asarray(cam14, np.float32)
        self.cam = cam14
        lib().orc_set_camera(self.h, _p(cam14[0:3].copy()), _p(cam14[3:6].copy()), _p(cam14[6:9].copy()),
                             _p(cam14[9:12].copy()), float(cam14[12]), float(cam14[13]))

    def trace_primary(self, offset, count, use_tree=False):
        tri = np.empty(count, np.int32); t = np.empty(count, np.float32)
        u = np.empty(count, np.float32); v = np.empty(count, np.float32)
        lib().orc_trace_primary(self.h, offset, count, int(use_tree), _p(tri), _p(t), _p(u), _p(v))
        return tri, t, u, v

    def view_rays(self, offset, count):
        out = np.empty((count, 6), np.float32)
        lib().orc_view_rays(self.h, offset, count, _p(out))
        return out

    def trace_rays(self, rays6, kind=0, use_tree=False):
        rays6 = np.ascontiguousarray(rays6, np.float32); n = len(rays6)
        tri = np.empty(n, np.int32); t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        lib().orc_trace_rays(self.h, _p(rays6), n, kind, int(use_tree), _p(tri), _p(t), _p(u), _p(v))
        return tri, t, u, v

    def trace_tex(self, rays6, ignore_backfaces=True):
        """Scene.trace, texcoord variant (scene.d:745-796): result (0 front, 1 back, 2 none), triangle, t, texcoords."""
        rays6 = np.ascontiguousarray(rays6, np.float32); n = len(rays6)
        res = np.empty(n, np.int32); tri = np.empty(n, np.int32); t = np.empty(n, np.float32); tc = np.empty((n, 2), np.float32)
        lib().orc_trace_tex(self.h, _p(rays6), n, int(ignore_backfaces), _p(res), _p(tri), _p(t), _p(tc))
        return res, tri, t, tc

    def interp_normal(self, tri, u, v):
        out = np.empty(3, np.float32)
        lib().orc_interp_normal(self.h, int(tri), float(u), float(v), _p(out))
        return out

    def frontface_disagreements(self, offset, count):
        return lib().orc_frontface_disagreements(self.h, offset, count)

    def light_groups(self):
        out = []
        for g in range(lib().orc_num_light_groups(self.h)):
            buf = np.zeros((self.T, 11), np.float32); area = C.c_float()
            n = lib().orc_light_group(self.h, g, _p(buf), self.T, C.byref(area))
            out.append((buf[:n].copy(), area.value))
        return out

    def check_tree(self):
        return lib().orc_check_tree(self.h)

    def bvh_count(self, nodes8, root_ref, tri_order, rays6, any_hit=False):
        """Box/triangle tests the kernels' traversal order performs on the product's flattened BVH."""
        nodes8 = np.ascontiguousarray(nodes8, np.float32); tri_order = np.ascontiguousarray(tri_order, np.uint32)
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        out = np.zeros(2, np.uint64); tri = np.empty(len(rays6), np.int32)
        lib().orc_bvh_count(self.h, _p(nodes8), int(root_ref), _p(tri_order), _p(rays6), len(rays6), int(any_hit), _p(out), _p(tri))
        return int(out[0]), int(out[1]), tri

    def render_mt(self, pixels, spp, nthreads=1, use_tree=True):
        """Reference-shaped renderer (bands + Mt19937).  Returns (seconds, counters[4])."""
        cn = np.zeros(4, np.uint64)
        sec = lib().orc_render_mt(self.h, _p(pixels), spp, nthreads, int(use_tree), _p(cn))
        return sec, cn

    def render_keyed(self, pixels, offset, spp, sample_base=0, nthreads=1, use_tree=False, sample_stride=1):
        cn = np.zeros(4, np.uint64)
        sec = lib().orc_render_keyed(self.h, offset, len(pixels), _p(pixels), spp, sample_base, sample_stride, nthreads,
                                     int(use_tree), _p(cn))
        return sec, cn


def default_sun_dir():
    v = np.array([1, 1, 3], np.float32)
    return (v * (np.float32(1.0) / np.sqrt(np.float32(11.0), dtype=np.float32))).astype(np.float32)


def default_sky_radiance():
    return (np.array([0.682, 0.977, 1.0], np.float32) * np.float32(0.1)).astype(np.float32)


def keyed_stream(seed, pixel, sample, stream, n):
    out = np.empty(n, np.uint32)
    lib().orc_keyed_stream(seed, pixel, sample, stream, n, _p(out))
    return out


def set_libm_trig(on):
    """Process-wide: the oracle calls this host's sinf / cosf instead of the fixed sequence S13 (for bounding what the
    choice changes over whole images; parity tests never set it)."""
    lib().orc_set_libm_trig(int(bool(on)))


def sincos(x):
    """S13: the oracle's fixed evaluation of sinf / cosf."""
    x = np.ascontiguousarray(x, np.float32); s = np.empty_like(x); c = np.empty_like(x)
    lib().orc_sincos(_p(x), len(x), _p(s), _p(c))
    return s, c


def pow_inv24(x):
    """S14: the oracle's fixed evaluation of powf(x, 1 / 2.4)."""
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    lib().orc_pow_inv24(_p(x), len(x), _p(y))
    return y


def angle_directions(phi, psi, normals):
    """angleToDirection (rendering.d:316-330); rows with a zero normal use angleToLocalDirection (:309-314)."""
    a = np.ascontiguousarray(np.column_stack([phi, psi, normals]), np.float32)
    out = np.empty((len(a), 3), np.float32)
    lib().orc_angle_directions(_p(a), len(a), _p(out))
    return out


def tonemap(pixels):
    rgb = np.empty((len(pixels), 3), np.uint8)
    lib().orc_tonemap(_p(pixels), len(pixels), _p(rgb))
    return rgb


def new_pixels(n):
    return np.zeros(n, PIXEL_DTYPE)
