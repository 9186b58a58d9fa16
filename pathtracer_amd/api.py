"""Host-side mirror of the reference's interface for its integrator + intersection path, on top of the
C ABI in include/ptgpu.h.  Names follow the reference where it has one:

  Camera(fov, aspectRatio), Camera.setTransform(from, to, up)        camera.d:16-29
  Scene(path, matFunc, modelMatrix), Scene.trace, Scene.hitsNothing  scene.d:193, 681, 801
  computeOutputColor(pixelOffset, pixels, gen)                        rendering.d:494-500
  Pixel {color, n, sum}                                               rendering.d:24-30
  drawScreen's tone map                                               main.d:82-95

The renderer is the GPU library; nothing here computes radiance or intersections on the CPU.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import PtConfig, PtStats

PIXEL_DTYPE = np.dtype([('color', '<f4', 3), ('n', '<f4'), ('sum', '<f4', 3)])      # rendering.d:24-30
MATERIAL_DTYPE = np.dtype([('emissive', '<f4'), ('color', '<f4', 3)])                # rendering.d:33-37

# material tables selectable in Scene(): rendering.d:247-275 and rendering.d:195-244
FILL_MATERIAL, FILL_MATERIAL_CITYMAP = 0, 1


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('ptgpu error %d: %s' % (code, msg))
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_sun_dir():
    """g_sunDir = vec3(1,1,3).normalized()  (rendering.d:72)"""
    v = np.array([1, 1, 3], np.float32)
    return (v * (np.float32(1.0) / np.sqrt(np.float32(11.0), dtype=np.float32))).astype(np.float32)


def default_sky_radiance():
    """g_skyRadiance = vec3(0.682, 0.977, 1.0) * 0.1  (rendering.d:74)"""
    return (np.array([0.682, 0.977, 1.0], np.float32) * np.float32(0.1)).astype(np.float32)


class Camera:
    """camera.d: the 14 floats of state are computed by the library (ptgpu_camera_look_at)."""

    def __init__(self, fov, aspectRatio):
        self.fov, self.aspect = float(fov), float(aspectRatio)
        self.state = None

    def setTransform(self, frm, to, up):
        out = np.zeros(14, np.float32)
        f, t, u = (np.ascontiguousarray(x, np.float32) for x in (frm, to, up))
        rc = _lib.load().ptgpu_camera_look_at(None, self.fov, self.aspect, _ptr(f), _ptr(t), _ptr(u), _ptr(out))
        if rc:
            raise PtError(rc, _lib.load().ptgpu_last_error().decode())
        self.state = out
        return self


class PathTracer:
    """One GPU's renderer: owns what the reference keeps in g_camera / g_scene / Pixel[]."""

    def __init__(self, width, height, device=0, **kw):
        self.lib = _lib.load()
        self.cfg = PtConfig()
        self.cfg.device, self.cfg.width, self.cfg.height = device, width, height
        self.cfg.hasLights = 1                                        # rendering.d:55
        self.cfg.sunDir[:] = [float(x) for x in default_sun_dir()]     # rendering.d:72-74
        self.cfg.sunRadiance[:] = [1.0, 0.984, 0.8]
        self.cfg.skyRadiance[:] = [float(x) for x in default_sky_radiance()]
        self._apply(kw)
        h = C.c_void_p()
        self._check(self.lib.ptgpu_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.width, self.height = width, height
        self._acc_keepalive = None

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, self.lib.ptgpu_last_error().decode(errors='replace'))

    _ALIASES = {'has_sky': 'hasSky', 'has_lights': 'hasLights', 'eps': 'floatEpsilon', 'rr': 'rrProb',
                'max_depth': 'maxDepth', 'n_indirect': 'numIndirect', 'sample_base': 'sampleBase', 'sample_stride': 'sampleStride',
                'batch_slots': 'batchSlots', 'max_leaf_tris': 'maxLeafTris', 'sun_dir': 'sunDir',
                'sun_radiance': 'sunRadiance', 'sky_radiance': 'skyRadiance'}

    def _apply(self, kw):
        for k, v in kw.items():
            k = self._ALIASES.get(k, k)
            if not hasattr(self.cfg, k):
                raise TypeError('unknown config field %r' % k)
            if k in ('sunDir', 'sunRadiance', 'skyRadiance'):
                getattr(self.cfg, k)[:] = [float(x) for x in v]
            elif k in ('floatEpsilon', 'rrProb'):
                setattr(self.cfg, k, float(v))
            else:
                setattr(self.cfg, k, int(v))

    def set_config(self, **kw):
        self._apply(kw)
        self._check(self.lib.ptgpu_set_config(self.h, C.byref(self.cfg)))

    def close(self):
        if getattr(self, 'h', None):
            self.lib.ptgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ scene + camera
    def Scene(self, path, matFunc=FILL_MATERIAL, modelMatrix=None):
        """New!Scene(path, matFunc, modelMatrix) (scene.d:193); modelMatrix None = raw mesh space."""
        mm = None if modelMatrix is None else np.ascontiguousarray(modelMatrix, np.float32).reshape(16)
        self._check(self.lib.ptgpu_load_thmodel(self.h, str(path).encode(), int(matFunc), _ptr(mm)))
        return self

    def upload_scene(self, v, n, mat_idx, materials):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
        n = np.ascontiguousarray(n, np.float32).reshape(-1, 9)
        mi = np.ascontiguousarray(mat_idx, np.uint32)
        m = np.ascontiguousarray(materials, np.float32).reshape(-1, 4)
        if not (len(v) == len(n) == len(mi)):
            raise ValueError('v, n and mat_idx must describe the same number of triangles')
        self._check(self.lib.ptgpu_upload_scene(self.h, _ptr(v), _ptr(n), _ptr(mi), _ptr(m), len(v), len(m)))
        return self

    def get_scene(self):
        st = self.stats()
        T = st.numTriangles
        v = np.zeros((T, 9), np.float32); n = np.zeros((T, 9), np.float32); mi = np.zeros(T, np.uint32)
        mats = np.zeros((256, 4), np.float32); nm = C.c_uint32()
        self._check(self.lib.ptgpu_get_scene(self.h, _ptr(v), _ptr(n), _ptr(mi), None, C.byref(nm)))
        mats = np.zeros((nm.value, 4), np.float32)
        self._check(self.lib.ptgpu_get_scene(self.h, None, None, None, _ptr(mats), None))
        return {'v': v.reshape(T, 3, 3), 'n': n.reshape(T, 3, 3), 'mat_idx': mi, 'materials': mats}

    def light_groups(self):
        """g_lightTriangles / g_totalLightSourceArea (rendering.d:47-48)."""
        out = []
        T = max(1, self.stats().numTriangles)
        for g in range(self.stats().numLightGroups):
            rows = np.zeros((T, 11), np.float32); cnt = C.c_uint32(); area = C.c_float()
            self._check(self.lib.ptgpu_get_light_group(self.h, g, _ptr(rows), T, C.byref(cnt), C.byref(area)))
            out.append((rows[:cnt.value].copy(), area.value))
        return out

    def get_bvh(self):
        st = self.stats()
        nodes = np.zeros((st.numNodes, 8), np.float32)
        order = np.zeros(max(1, st.numTriangles), np.uint32)
        self._check(self.lib.ptgpu_get_bvh(self.h, _ptr(nodes), _ptr(order)))
        return nodes, order

    def set_camera(self, cam):
        """cam: Camera (after setTransform) or 14 floats pos dir right up tanFov aspect (camera.d:11-12)."""
        s = np.ascontiguousarray(cam.state if isinstance(cam, Camera) else cam, np.float32)
        self._check(self.lib.ptgpu_set_camera(self.h, _ptr(s[0:3].copy()), _ptr(s[3:6].copy()), _ptr(s[6:9].copy()),
                                              _ptr(s[9:12].copy()), float(s[12]), float(s[13])))
        return self

    # ------------------------------------------------------------------ rendering
    def computeOutputColor(self, pixelOffset, pixelCount, spp=1):
        """rendering.d:494-500 for a whole window of pixels, spp passes (asynchronous)."""
        self._check(self.lib.ptgpu_render(self.h, pixelOffset, pixelCount, spp))

    render = computeOutputColor

    def render_noalloc(self, pixelOffset, pixelCount, spp=1):
        """computeOutputColor for callers that cannot allocate: one batch in the library's static scratch buffer."""
        self._check(self.lib.ptgpu_render_noalloc(self.h, pixelOffset, pixelCount, spp))

    def render_frame(self, spp):
        self.computeOutputColor(0, self.width * self.height, spp)

    def render_bands(self, band_rows, band_first, band_stride, spp):
        """main.d:184-192's 4-scanline tasks, every band_stride-th band starting at band_first."""
        self._check(self.lib.ptgpu_render_bands(self.h, band_rows, band_first, band_stride, spp))

    def synchronize(self):
        self._check(self.lib.ptgpu_synchronize(self.h))

    def read_pixels(self, offset=0, count=None):
        count = self.width * self.height - offset if count is None else count
        out = np.zeros(count, PIXEL_DTYPE)
        self._check(self.lib.ptgpu_read_pixels(self.h, offset, count, _ptr(out)))
        return out

    def write_pixels(self, pixels, offset=0):
        pixels = np.ascontiguousarray(pixels, PIXEL_DTYPE)
        self._check(self.lib.ptgpu_write_pixels(self.h, offset, len(pixels), _ptr(pixels)))

    def clear_pixels(self):
        self._check(self.lib.ptgpu_clear_pixels(self.h))

    def tonemap(self, offset=0, count=None):
        count = self.width * self.height - offset if count is None else count
        out = np.zeros((count, 3), np.uint8)
        self._check(self.lib.ptgpu_tonemap(self.h, offset, count, _ptr(out)))
        return out

    def tonemap_begin(self, out, offset=0):
        """First half of tonemap(): queues the tone map and its copy into `out` (uint8, count x 3, kept alive by the
        caller) and returns; tonemap_end() waits for the bytes.  Passes rendered in between overlap the copy."""
        assert out.dtype == np.uint8 and out.flags['C_CONTIGUOUS'] and out.size % 3 == 0
        self._check(self.lib.ptgpu_tonemap_begin(self.h, offset, out.size // 3, _ptr(out)))

    def tonemap_end(self):
        self._check(self.lib.ptgpu_tonemap_end(self.h))

    def host_array(self, shape, dtype):
        """A numpy array in page-locked memory of the library (frame transfers copy straight into / out of it).
        Freed by host_free(array); must not be used afterwards."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self.lib.ptgpu_host_alloc(self.h, max(1, n), C.byref(p)))
        buf = (C.c_uint8 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        arr[...] = np.zeros((), dtype)
        self._host_ptrs = getattr(self, '_host_ptrs', {}); self._host_ptrs[arr.ctypes.data] = p.value
        return arr

    def host_free(self, array):
        p = self._host_ptrs.pop(array.ctypes.data)
        self._check(self.lib.ptgpu_host_free(self.h, C.c_void_p(p)))

    # ------------------------------------------------------------------ Scene.trace / Scene.hitsNothing
    def trace(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6); n = len(rays6)
        tri = np.zeros(n, np.int32); t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32)
        self._check(self.lib.ptgpu_scene_trace(self.h, _ptr(rays6), n, _ptr(tri), _ptr(t), _ptr(u), _ptr(v), _ptr(nrm)))
        return tri, t, u, v, nrm

    def trace_tex(self, rays6, ignore_backfaces=True):
        """Scene.trace(ray, rayPos, texcoords, data, ignoreBackfaces) (scene.d:745-796) ->
        (TraceResult 0 front / 1 back / 2 none, triangle id, rayPos, texcoords)."""
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6); n = len(rays6)
        res = np.zeros(n, np.int32); tri = np.zeros(n, np.int32); t = np.zeros(n, np.float32)
        tc = np.zeros((n, 2), np.float32)
        self._check(self.lib.ptgpu_scene_trace_tex(self.h, _ptr(rays6), n, 1 if ignore_backfaces else 0, _ptr(res),
                                                   _ptr(tri), _ptr(t), _ptr(tc)))
        return res, tri, t, tc

    def set_texcoords(self, tex):
        """TriangleData.tex (scene.d:305-307): T x 3 x (s, t) in triangle input order."""
        tex = np.ascontiguousarray(tex, np.float32).reshape(-1, 6)
        if len(tex) != self.stats().numTriangles:
            raise ValueError('tex must hold 3 x 2 floats for every triangle of the scene')
        self._check(self.lib.ptgpu_set_texcoords(self.h, _ptr(tex)))
        return self

    def get_texcoords(self):
        tex = np.zeros((max(1, self.stats().numTriangles), 6), np.float32)
        self._check(self.lib.ptgpu_get_texcoords(self.h, _ptr(tex)))
        return tex[:self.stats().numTriangles].reshape(-1, 3, 2)

    def hitsNothing(self, rays6):
        rays6 = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6); n = len(rays6)
        out = np.zeros(n, np.uint8)
        self._check(self.lib.ptgpu_scene_hits_nothing(self.h, _ptr(rays6), n, _ptr(out)))
        return out.astype(bool)

    def trace_primary(self, offset=0, count=None):
        count = self.width * self.height - offset if count is None else count
        tri = np.zeros(count, np.int32); t = np.zeros(count, np.float32)
        u = np.zeros(count, np.float32); v = np.zeros(count, np.float32)
        self._check(self.lib.ptgpu_trace_primary(self.h, offset, count, _ptr(tri), _ptr(t), _ptr(u), _ptr(v)))
        return tri, t, u, v

    def view_rays(self, offset=0, count=None):
        count = self.width * self.height - offset if count is None else count
        out = np.zeros((count, 6), np.float32)
        self._check(self.lib.ptgpu_view_rays(self.h, offset, count, _ptr(out)))
        return out

    def rng_stream(self, pixel, sample, stream, n):
        out = np.zeros(n, np.uint32)
        self._check(self.lib.ptgpu_rng_stream(self.h, pixel, sample, stream, n, _ptr(out)))
        return out

    def angle_directions(self, phi, psi, normals):
        """angleToDirection(phi, psi, normal) (rendering.d:316-330) on the device; zero normal rows:
        angleToLocalDirection (rendering.d:309-314)."""
        a = np.ascontiguousarray(np.column_stack([phi, psi, normals]), np.float32)
        out = np.zeros((len(a), 3), np.float32)
        self._check(self.lib.ptgpu_angle_directions(self.h, _ptr(a), len(a), _ptr(out)))
        return out

    def coverage_mask(self, constant_sky=False):
        """Boolean array (height*width): pixels whose view rays may hit a triangle; None when no mask can be built
        for the current camera (rendering then culls by the scene's bounding box).  constant_sky=True returns
        (covered, constantSky): the second marks pixels all of whose samples are the sky constant."""
        npix = self.width * self.height
        bits = np.zeros((npix + 31) // 32, np.uint32); sky = np.zeros_like(bits); usable = np.zeros(1, np.uint32)
        self._check(self.lib.ptgpu_get_coverage_mask(self.h, _ptr(bits), _ptr(sky), _ptr(usable)))
        if not usable[0]:
            return (None, None) if constant_sky else None
        unpack = lambda w: np.unpackbits(w.view(np.uint8), bitorder='little')[:npix].astype(bool)
        return (unpack(bits), unpack(sky)) if constant_sky else unpack(bits)

    # ------------------------------------------------------------------ statistics
    def stats(self):
        st = PtStats()
        self._check(self.lib.ptgpu_get_stats(self.h, C.byref(st)))
        return st

    def reset_stats(self):
        self._check(self.lib.ptgpu_reset_stats(self.h))

    # ------------------------------------------------------------------ device-memory plumbing (torch)
    def bind_accumulator(self, tensor):
        """tensor: contiguous float32 CUDA tensor with width*height*4 elements (sum.rgb, n)."""
        nbytes = tensor.numel() * tensor.element_size()
        self._check(self.lib.ptgpu_bind_accumulator(self.h, C.c_void_p(tensor.data_ptr()), nbytes))
        self._acc_keepalive = tensor

    def set_stream(self, stream_ptr):
        self._check(self.lib.ptgpu_set_stream(self.h, C.c_void_p(stream_ptr)))


class _Borrowed(PathTracer):
    """A PathTracer view of a handle that a GpuGroup owns (never destroyed from here)."""

    def __init__(self, lib, handle, width, height):
        self.lib, self.h, self.width, self.height = lib, handle, width, height
        self.cfg = PtConfig(); self._acc_keepalive = None

    def close(self):
        self.h = None


class GpuGroup:
    """Several GPUs of one node as one renderer (ptgpu_group_*): main.d:184-215's bands, a GPU per worker thread."""

    def __init__(self, width, height, devices, root=0, **kw):
        self.lib = _lib.load()
        proto = PathTracer.__new__(PathTracer)
        proto.cfg = PtConfig(); proto.cfg.width, proto.cfg.height, proto.cfg.hasLights = width, height, 1
        proto.cfg.sunDir[:] = [float(x) for x in default_sun_dir()]
        proto.cfg.sunRadiance[:] = [1.0, 0.984, 0.8]
        proto.cfg.skyRadiance[:] = [float(x) for x in default_sky_radiance()]
        proto._apply(kw)
        proto.h = None
        devs = np.ascontiguousarray(devices, np.int32)
        g = C.c_void_p()
        self._check(self.lib.ptgpu_group_create(C.byref(proto.cfg), _ptr(devs), len(devs), root, C.byref(g)))
        self.g, self.width, self.height, self.n = g, width, height, len(devs)

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, self.lib.ptgpu_last_error().decode(errors='replace'))

    def close(self):
        if getattr(self, 'g', None):
            self.lib.ptgpu_group_destroy(self.g); self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, root, rccl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.lib.ptgpu_group_info(self.g, C.byref(n), C.byref(root), C.byref(rccl)))
        return {'devices': n.value, 'root': root.value, 'rccl': bool(rccl.value)}

    def rank(self, i):
        h = C.c_void_p(); self._check(self.lib.ptgpu_group_handle(self.g, i, C.byref(h)))
        return _Borrowed(self.lib, h, self.width, self.height)

    def frame(self):
        h = C.c_void_p(); self._check(self.lib.ptgpu_group_frame(self.g, C.byref(h)))
        return _Borrowed(self.lib, h, self.width, self.height)

    def Scene(self, path, matFunc=FILL_MATERIAL, modelMatrix=None):
        mm = None if modelMatrix is None else np.ascontiguousarray(modelMatrix, np.float32).reshape(16)
        self._check(self.lib.ptgpu_group_load_thmodel(self.g, str(path).encode(), int(matFunc), _ptr(mm)))
        return self

    def set_camera(self, cam):
        s = np.ascontiguousarray(cam.state if isinstance(cam, Camera) else cam, np.float32)
        self._check(self.lib.ptgpu_group_set_camera(self.g, _ptr(s[0:3].copy()), _ptr(s[3:6].copy()), _ptr(s[6:9].copy()),
                                                    _ptr(s[9:12].copy()), float(s[12]), float(s[13])))
        return self

    def render(self, spp, split='bands', band_rows=4):
        self._check(self.lib.ptgpu_group_render(self.g, {'bands': 0, 'samples': 1}[split], band_rows, spp))

    def combine(self):
        self._check(self.lib.ptgpu_group_combine(self.g))
        return self.frame()

    def clear(self):
        self._check(self.lib.ptgpu_group_clear(self.g))


def band_rows(height, band_rows_, first, stride):
    """ptgpu_band_rows: the scanlines of bands first, first + stride, ... (main.d:184-192)."""
    lib = _lib.load()
    cnt = C.c_uint32()
    rc = lib.ptgpu_band_rows(height, band_rows_, first, stride, None, 0, C.byref(cnt))
    if rc:
        raise PtError(rc, lib.ptgpu_last_error().decode())
    rows = np.zeros(max(1, cnt.value), np.uint32)
    lib.ptgpu_band_rows(height, band_rows_, first, stride, _ptr(rows), cnt.value, C.byref(cnt))
    return rows[:cnt.value]


def pixels_from_accumulator(acc):
    """(sum.rgb, n) rows -> Pixel records with color = sum / n (rendering.d:483)."""
    acc = np.asarray(acc, np.float32).reshape(-1, 4)
    out = np.zeros(len(acc), PIXEL_DTYPE)
    out['sum'] = acc[:, :3]; out['n'] = acc[:, 3]
    nz = acc[:, 3] > 0
    out['color'][nz] = acc[nz, :3] / acc[nz, 3:4]
    return out
