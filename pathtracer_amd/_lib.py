"""ctypes binding of libptgpu.so (include/ptgpu.h).  There is no CPU fallback: if the library is
missing or no GPU is present the calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PTGPU_LIBRARY') or os.path.join(_HERE, 'libptgpu.so')   # override: A/B runs of two builds

PT_OK, PT_E_INVALID, PT_E_STATE, PT_E_DEVICE, PT_E_IO, PT_E_NOMEM = 0, -1, -2, -3, -4, -5
PT_FLAG_COUNT_VISITS, PT_FLAG_NO_LDS_SCENE, PT_FLAG_TIME_KERNELS, PT_FLAG_NO_COVERAGE_MASK = 1, 2, 4, 8
PT_FLAG_GPU_BVH, PT_FLAG_HOST_BVH, PT_FLAG_NO_GRAPH, PT_FLAG_NO_FUSE, PT_FLAG_QUEUED_VIEW_RAYS, PT_FLAG_NO_TREELETS, PT_FLAG_STATIC_RUNS = 16, 32, 64, 128, 256, 1024, 512
PT_FLAG_NO_PIXEL_SUMS = 2048
PT_FLAG_NO_HIT_ORDER = 4096
PT_FLAG_HIT_ORDER = 8192
PT_K_NAMES = ('extend', 'connect', 'occluded', 'shade', 'accumulate', 'other', 'raygen', 'raygen_extend', 'shade_occluded',
              'shade_traced', 'sample_traced', 'pixel_traced')


class PtConfig(C.Structure):
    _fields_ = [('device', C.c_int32), ('width', C.c_uint32), ('height', C.c_uint32), ('mode', C.c_uint32),
                ('hasSky', C.c_uint32), ('hasLights', C.c_uint32), ('sunDir', C.c_float * 3),
                ('sunRadiance', C.c_float * 3), ('skyRadiance', C.c_float * 3), ('floatEpsilon', C.c_float),
                ('rrProb', C.c_float), ('maxDepth', C.c_uint32), ('numIndirect', C.c_uint32), ('seed', C.c_uint32),
                ('sampleBase', C.c_uint32), ('sampleStride', C.c_uint32), ('batchSlots', C.c_uint32), ('maxLeafTris', C.c_uint32),
                ('flags', C.c_uint32)]


class PtStats(C.Structure):
    _fields_ = [('raysPrimary', C.c_uint64), ('raysLight', C.c_uint64), ('raysIndirect', C.c_uint64),
                ('raysOcclusion', C.c_uint64), ('samples', C.c_uint64), ('nodesClosest', C.c_uint64),
                ('trisClosest', C.c_uint64), ('nodesAny', C.c_uint64), ('trisAny', C.c_uint64), ('raysCulled', C.c_uint64),
                ('hits', C.c_uint64), ('envEntries', C.c_uint64),
                ('hitsPrimary', C.c_uint64), ('raysPrimaryInShade', C.c_uint64), ('raysLightInShade', C.c_uint64), ('raysIndirectInShade', C.c_uint64), ('raysOcclusionInShade', C.c_uint64),
                ('raysLightInSample', C.c_uint64), ('raysIndirectInSample', C.c_uint64), ('raysOcclusionInSample', C.c_uint64),
                ('chainHitsFromSample', C.c_uint64), ('samplesConstantSky', C.c_uint64), ('sceneBytes', C.c_uint64),
                ('kernelMs', C.c_double * 12), ('kernelLaunches', C.c_uint64 * 12), ('renderMs', C.c_double),
                ('numTriangles', C.c_uint32), ('numNodes', C.c_uint32), ('bvhDepth', C.c_uint32),
                ('numLightGroups', C.c_uint32), ('sceneInLds', C.c_uint32), ('bvhOnGpu', C.c_uint32),
                ('bvhBuildSeconds', C.c_double), ('maskBuildMs', C.c_double)]


# every symbol include/ptgpu.h declares: name -> (restype, argtypes)
_VP, _U32, _I, _F = C.c_void_p, C.c_uint32, C.c_int, C.c_float
SYMBOLS = {
    'ptgpu_last_error': (C.c_char_p, []),
    'ptgpu_create': (_I, [C.POINTER(PtConfig), C.POINTER(_VP)]),
    'ptgpu_destroy': (_I, [_VP]),
    'ptgpu_set_config': (_I, [_VP, C.POINTER(PtConfig)]),
    'ptgpu_upload_scene': (_I, [_VP, _VP, _VP, _VP, _VP, _U32, _U32]),
    'ptgpu_load_thmodel': (_I, [_VP, C.c_char_p, _I, _VP]),
    'ptgpu_get_scene': (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'ptgpu_get_light_group': (_I, [_VP, _U32, _VP, _U32, _VP, _VP]),
    'ptgpu_get_bvh': (_I, [_VP, _VP, _VP]),
    'ptgpu_set_camera': (_I, [_VP, _VP, _VP, _VP, _VP, _F, _F]),
    'ptgpu_camera_look_at': (_I, [_VP, _F, _F, _VP, _VP, _VP, _VP]),
    'ptgpu_render': (_I, [_VP, _U32, _U32, _U32]),
    'ptgpu_render_noalloc': (_I, [_VP, _U32, _U32, _U32]),
    'ptgpu_render_bands': (_I, [_VP, _U32, _U32, _U32, _U32]),
    'ptgpu_synchronize': (_I, [_VP]),
    'ptgpu_read_pixels': (_I, [_VP, _U32, _U32, _VP]),
    'ptgpu_write_pixels': (_I, [_VP, _U32, _U32, _VP]),
    'ptgpu_clear_pixels': (_I, [_VP]),
    'ptgpu_tonemap': (_I, [_VP, _U32, _U32, _VP]),
    'ptgpu_tonemap_begin': (_I, [_VP, _U32, _U32, _VP]),
    'ptgpu_tonemap_end': (_I, [_VP]),
    'ptgpu_host_alloc': (_I, [_VP, C.c_uint64, C.POINTER(_VP)]),
    'ptgpu_host_free': (_I, [_VP, _VP]),
    'ptgpu_scene_trace': (_I, [_VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP]),
    'ptgpu_scene_trace_tex': (_I, [_VP, _VP, _U32, _I, _VP, _VP, _VP, _VP]),
    'ptgpu_set_texcoords': (_I, [_VP, _VP]),
    'ptgpu_get_texcoords': (_I, [_VP, _VP]),
    'ptgpu_scene_hits_nothing': (_I, [_VP, _VP, _U32, _VP]),
    'ptgpu_trace_primary': (_I, [_VP, _U32, _U32, _VP, _VP, _VP, _VP]),
    'ptgpu_view_rays': (_I, [_VP, _U32, _U32, _VP]),
    'ptgpu_rng_stream': (_I, [_VP, _U32, _U32, _U32, _U32, _VP]),
    'ptgpu_angle_directions': (_I, [_VP, _VP, _U32, _VP]),
    'ptgpu_get_coverage_mask': (_I, [_VP, _VP, _VP, _VP]),
    'ptgpu_get_stats': (_I, [_VP, C.POINTER(PtStats)]),
    'ptgpu_reset_stats': (_I, [_VP]),
    'ptgpu_bind_accumulator': (_I, [_VP, _VP, C.c_uint64]),
    'ptgpu_get_accumulator': (_I, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    'ptgpu_set_stream': (_I, [_VP, _VP]),
}

SYMBOLS.update({
    'ptgpu_group_create': (_I, [C.POINTER(PtConfig), _VP, _U32, _U32, C.POINTER(_VP)]),
    'ptgpu_group_destroy': (_I, [_VP]),
    'ptgpu_group_info': (_I, [_VP, _VP, _VP, _VP]),
    'ptgpu_group_handle': (_I, [_VP, _U32, C.POINTER(_VP)]),
    'ptgpu_group_frame': (_I, [_VP, C.POINTER(_VP)]),
    'ptgpu_group_load_thmodel': (_I, [_VP, C.c_char_p, _I, _VP]),
    'ptgpu_group_upload_scene': (_I, [_VP, _VP, _VP, _VP, _VP, _U32, _U32]),
    'ptgpu_group_set_camera': (_I, [_VP, _VP, _VP, _VP, _VP, _F, _F]),
    'ptgpu_group_camera_look_at': (_I, [_VP, _F, _F, _VP, _VP, _VP]),
    'ptgpu_group_render': (_I, [_VP, _I, _U32, _U32]),
    'ptgpu_group_combine': (_I, [_VP]),
    'ptgpu_group_clear': (_I, [_VP]),
    'ptgpu_band_rows': (_I, [_U32, _U32, _U32, _U32, _VP, _U32, _VP]),
})
PT_SPLIT_BANDS, PT_SPLIT_SAMPLES = 0, 1

_lib = None


def load():
    """Loads libptgpu.so and binds every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('%s is missing: build it with `make -C pathtracer_amd/csrc` '
                               '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
