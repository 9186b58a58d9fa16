"""ptgpu_render_noalloc (the entry point that allocates nothing: wavefront state in a static device array) against
ptgpu_render through the C ABI: bit-identical accumulators on two calls in a row, the second with a smaller window;
device memory in use does not change across the calls; requests beyond the array are refused."""
import numpy as np
import pytest

import pathtracer_amd as P
from conftest import make_tracer

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


@pytest.mark.parametrize('name,sky,mode,W,H,spp', [('cornell-box', 0, 0, 256, 256, 3), ('teapot', 1, 0, 320, 240, 4),
                                                   ('chest1', 1, 1, 128, 96, 2)])
def test_noalloc_render_equals_render_on_two_calls_in_a_row(require_gpu, name, sky, mode, W, H, spp):
    ref = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=9)
    na = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=9)
    ref.render(0, W * H, spp)
    na.render_noalloc(0, W * H, spp)                                   # first call: the whole frame
    a, b = ref.read_pixels(), na.read_pixels()
    assert (b['n'] == spp).all()
    assert np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32)) and np.array_equal(a['n'], b['n'])
    # second call, smaller: a window that starts and ends inside scanlines, one sample per pixel more
    off, cnt = 17 * W + 5, 23 * W + 11
    ref.render(off, cnt, 1)
    ref.synchronize(); na.synchronize(); before = _free_bytes()        # (a stream's first launches and copies make the HIP
    na.render_noalloc(off, cnt, 1)                                     # runtime set up memory of its own: measured once warm)
    na.synchronize()
    assert _free_bytes() == before                                     # nothing was allocated on the device
    a, b = ref.read_pixels(), na.read_pixels()
    assert np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32)) and np.array_equal(a['n'], b['n'])
    assert (b['n'][off:off + cnt] == spp + 1).all() and (b['n'][:off] == spp).all()
    # a third, on a handle whose masks exist (ptgpu_render built them): same bits again
    ref.render_noalloc(0, 40 * W, 1); na.render(0, 40 * W, 1)
    a, b = ref.read_pixels(), na.read_pixels()
    assert np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32))


def test_noalloc_limits_and_errors(require_gpu):
    W = H = 1024
    pt = make_tracer('cornell-box', W, H, seed=1)
    with pytest.raises(P.PtError) as e:
        pt.render_noalloc(0, W * H, 1)                                 # 1 Mi samples: beyond the array
    assert e.value.code == -1 and 'exceeds the 458752 samples' in str(e.value) and '96 MiB' in str(e.value)
    pt.render_noalloc(0, 458752, 1); pt.synchronize()                  # the stated maximum works
    assert (pt.read_pixels()['n'][:458752] == 1).all() and (pt.read_pixels()['n'][458752:] == 0).all()
    pt.render_noalloc(0, 0, 1)                                         # empty window: nothing to do
    with pytest.raises(P.PtError):
        pt.render_noalloc(W * H - 10, 11, 1)                           # window beyond the frame
    m1 = make_tracer('chest1', 512, 512, has_sky=1, mode=1, seed=1)
    with pytest.raises(P.PtError) as e:
        m1.render_noalloc(0, 512 * 512, 1)
    assert 'exceeds the 65536 samples' in str(e.value)
    fresh = P.PathTracer(64, 64, device=0)
    with pytest.raises(P.PtError) as e:
        fresh.render_noalloc(0, 16, 1)
    assert e.value.code == -2                                          # no scene yet
