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
    W = H = 1536
    pt = make_tracer('cornell-box', W, H, seed=1)
    ref = make_tracer('cornell-box', W, H, seed=1)
    with pytest.raises(P.PtError) as e:
        pt.render_noalloc(0, W * H, 1)                                 # 2.25 Mi samples: beyond the array
    assert e.value.code == -1 and 'exceeds the 1769472 samples' in str(e.value) and '96 MiB' in str(e.value)
    pt.render_noalloc(0, 1769472, 1); ref.render(0, 1769472, 1)        # the stated maximum works
    a, b = ref.read_pixels(), pt.read_pixels()
    assert (b['n'][:1769472] == 1).all() and (b['n'][1769472:] == 0).all()
    assert np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32))
    pt.render_noalloc(0, 0, 1)                                         # empty window: nothing to do
    with pytest.raises(P.PtError):
        pt.render_noalloc(W * H - 10, 11, 1)                           # window beyond the frame
    # a scene whose stages are separate launches keeps its queues in the array as well: less room for samples
    q = make_tracer('cornell-box', 1024, 1024, seed=1, flags=P.PT_FLAG_NO_FUSE)
    with pytest.raises(P.PtError) as e:
        q.render_noalloc(0, 1024 * 1024, 1)
    assert 'exceeds the 458752 samples' in str(e.value)
    m1 = make_tracer('chest1', 512, 512, has_sky=1, mode=1, seed=1)
    with pytest.raises(P.PtError) as e:
        m1.render_noalloc(0, 512 * 512, 1)
    assert 'exceeds the 65536 samples' in str(e.value)
    m1.render_noalloc(0, 65536, 1)
    r1 = make_tracer('chest1', 512, 512, has_sky=1, mode=1, seed=1)
    r1.render(0, 65536, 1)
    a, b = r1.read_pixels(), m1.read_pixels()
    assert np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32)) and np.array_equal(a['n'], b['n'])
    t1 = make_tracer('teapot', 512, 512, has_sky=1, mode=1, seed=1)
    with pytest.raises(P.PtError) as e:
        t1.render_noalloc(0, 512 * 512, 1)
    assert 'exceeds the 65536 samples' in str(e.value)
    fresh = P.PathTracer(64, 64, device=0)
    with pytest.raises(P.PtError) as e:
        fresh.render_noalloc(0, 16, 1)
    assert e.value.code == -2                                          # no scene yet


def test_noalloc_calls_on_one_gpu_take_turns(require_gpu):
    """Every handle and stream of a GPU shares that GPU's scratch array.  Calls issued without any synchronisation in
    between -- from two handles with streams of their own, in one thread and from two threads at once -- are put in
    order by the library (each starts behind the last kernel of the one before), so each handle's image equals the
    image it renders alone."""
    import threading
    specs = [('cornell-box', 256, 256, dict(seed=5)), ('chest1', 160, 120, dict(has_sky=1, mode=1, seed=6))]
    solo = []
    for name, W, H, cfg in specs:
        pt = make_tracer(name, W, H, **cfg)
        for _ in range(4):
            pt.render(0, W * H, 1)
        solo.append(pt.read_pixels()); del pt
    for threaded in (False, True):
        hs = [make_tracer(name, W, H, **cfg) for name, W, H, cfg in specs]
        if not threaded:
            for _ in range(4):
                for pt, (name, W, H, cfg) in zip(hs, specs):
                    pt.render_noalloc(0, W * H, 1)                     # no synchronize between the two handles' calls
        else:
            errors = []
            def work(pt, n):
                try:
                    for _ in range(4):
                        pt.render_noalloc(0, n, 1)
                except Exception as e:
                    errors.append(e)
            th = [threading.Thread(target=work, args=(pt, W * H)) for pt, (name, W, H, cfg) in zip(hs, specs)]
            for t in th: t.start()
            for t in th: t.join()
            assert not errors, errors
        for pt, ref in zip(hs, solo):
            got = pt.read_pixels()
            assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
