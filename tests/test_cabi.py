"""The C-ABI library loads, exports every symbol include/ptgpu.h declares, and reports errors by code +
ptgpu_last_error() instead of throwing (the reference throws RCException, modelloader.d:134, scene.d:1007).
No compute call is made here (no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import _lib
from conftest import ROOT, model_path


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ptgpu.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ptgpu_\w+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    decl = declared_symbols()
    assert len(decl) >= 25
    assert sorted(_lib.SYMBOLS) == decl                     # the ctypes table is the header, nothing more or less
    lib = C.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.PtConfig) == 4 * 6 + 12 * 3 + 4 * 2 + 4 * 8          # all 4-byte fields, no padding
    assert P.PIXEL_DTYPE.itemsize == 28 and P.MATERIAL_DTYPE.itemsize == 16   # rendering.d:24-37
    st = _lib.PtStats()
    assert C.sizeof(st) == 23 * 8 + 12 * 8 + 12 * 8 + 8 + 6 * 4 + 8 + 8


def test_host_only_handle_fails_loudly_on_gpu_calls():
    pt = P.PathTracer(16, 16, device=-1)
    pt.Scene(model_path('chest1'))
    pt.set_camera(P.Camera(30.0, 1.0).setTransform((3, 3, 3), (0, 0, 0), (0, 0, 1)))
    for call in (lambda: pt.render_frame(1), lambda: pt.read_pixels(), lambda: pt.trace_primary(),
                 lambda: pt.trace(np.zeros((1, 6), np.float32)), lambda: pt.hitsNothing(np.zeros((1, 6), np.float32)),
                 lambda: pt.tonemap(), lambda: pt.view_rays(), lambda: pt.rng_stream(0, 0, 0, 4),
                 lambda: pt.render_bands(4, 0, 1, 1)):
        with pytest.raises(P.PtError) as e:
            call()
        assert e.value.code == _lib.PT_E_DEVICE and 'without a device' in str(e.value)


def test_argument_errors():
    with pytest.raises(P.PtError) as e:
        P.PathTracer(0, 16, device=-1)
    assert e.value.code == _lib.PT_E_INVALID
    with pytest.raises(P.PtError):
        P.PathTracer(16, 16, device=-1, mode=2)
    pt = P.PathTracer(16, 16, device=-1)
    lib = _lib.load()
    assert lib.ptgpu_render(None, 0, 1, 1) == _lib.PT_E_INVALID
    assert b'null' in lib.ptgpu_last_error()
    assert lib.ptgpu_render(pt.h, 0, 16 * 16 + 1, 1) == _lib.PT_E_INVALID       # window beyond the frame
    assert b'exceeds the frame' in lib.ptgpu_last_error()
    assert lib.ptgpu_render_bands(pt.h, 0, 0, 1, 1) == _lib.PT_E_INVALID
    assert lib.ptgpu_render_bands(pt.h, 4, 2, 2, 1) == _lib.PT_E_INVALID        # bandFirst >= bandStride
    v = np.zeros((1, 9), np.float32); mi = np.array([3], np.uint32); m = np.zeros((1, 4), np.float32)
    with pytest.raises(P.PtError) as e:
        pt.upload_scene(v, v, mi, m)                                             # material index out of range
    assert e.value.code == _lib.PT_E_INVALID
    with pytest.raises(P.PtError) as e:
        pt.set_config(width=32)
    assert 'fixed at ptgpu_create' in str(e.value)
    assert lib.ptgpu_destroy(None) == 0


def test_call_order_errors():
    pt = P.PathTracer(16, 16, device=-1)
    lib = _lib.load()
    assert lib.ptgpu_get_bvh(pt.h, None, None) == _lib.PT_E_STATE                # no scene yet
    assert lib.ptgpu_get_light_group(pt.h, 0, None, 0, None, None) == _lib.PT_E_STATE
    pt.Scene(model_path('cornell-box'))
    assert lib.ptgpu_get_light_group(pt.h, 1, None, 0, None, None) == _lib.PT_E_INVALID


def test_too_many_emissive_materials_rejected():
    pt = P.PathTracer(16, 16, device=-1)
    T = 12
    v = np.random.default_rng(0).normal(size=(T, 9)).astype(np.float32)
    mats = np.tile(np.array([[1.0, 0.5, 0.5, 0.5]], np.float32), (T, 1))
    with pytest.raises(P.PtError):
        pt.upload_scene(v, v, np.arange(T, dtype=np.uint32), mats)


def test_d_binding_text_follows_the_header():
    """INTEGRATION.md's D module is what scripts/gen_d_binding.py makes of include/ptgpu.h: every constant, struct and
    function of the header, nothing else."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_d_binding', os.path.join(ROOT, 'scripts', 'gen_d_binding.py'))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    _, m = gen.integration_block()
    assert m, 'INTEGRATION.md has no D block'
    assert m.group(1) == gen.generate(), 'run `python3 scripts/gen_d_binding.py --update` after changing include/ptgpu.h'
    dtext = m.group(1)
    for name in declared_symbols():
        assert re.search(r'\b%s\(' % name, dtext), name
    for struct in ('PtConfig', 'PtStats', 'PtPixel', 'PtMaterial', 'PtHandle', 'PtGroup'):
        assert 'struct ' + struct in dtext
    assert dtext.count('sampleStride') == 1 and 'double maskBuildMs;' in dtext
