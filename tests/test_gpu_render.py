"""GPU parity of the integrator (rendering.d:278-500) through the C ABI.

Two kinds of comparison with the CPU oracle:
  * same keyed RNG streams on both sides.  Mode 0 (the reference's HEAD) is bit-identical, sky and lights included:
    every operation, sin / cos among them, has one fixed evaluation on both sides, and radiance is summed in the
    source's order.  Mode 1: every ray, hit and visibility decision is identical (ray counts equal exactly); radiance
    agrees to a few ulp, because the kernels carry a chain's throughput forward (T * color * L) where the recursion
    multiplies on the way back (color * (Ld + color * (...))), and float products do not associate;
  * the reference-shaped oracle (Mt19937 per 4-scanline band, sphere-tree traversal): relative RMSE
    sqrt(mean((I_gpu - I_ref)^2)) / mean(I_ref) within 2x the RMSE between two oracle runs with different seeds
    at the same spp, and box-filtered bias within 1 %.
"""
import os

import numpy as np
import pytest

import oracle_py
import pathtracer_amd as P
from conftest import CAMERAS, GOLDEN, make_oracle, make_tracer, soup

pytestmark = pytest.mark.gpu
NAMES = sorted(CAMERAS)


def rel_rmse(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / b.mean())


def box(img, H, W, k=8):
    x = img.reshape(H // k, k, W // k, k, 3).astype(np.float64)
    return x.mean((1, 3))


@pytest.mark.parametrize('name', NAMES)
def test_image_equals_oracle_with_same_streams_mode0(require_gpu, name, flats):
    W = H = 96
    sky = name != 'cornell-box'
    pt = make_tracer(name, W, H, has_sky=int(sky), seed=99)
    o = make_oracle(flats[name], name, W, H, has_sky=sky, seed=99)
    pt.render_frame(16)
    got = pt.read_pixels()
    ref = oracle_py.new_pixels(W * H); o.render_keyed(ref, 0, 16, nthreads=8)
    assert np.array_equal(got['n'], ref['n']) and (got['n'] == 16).all()
    assert np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))           # bit-identical
    assert np.array_equal(got['color'].view(np.uint32), ref['color'].view(np.uint32))
    st = pt.stats()
    assert st.samples == W * H * 16 and st.raysPrimary == W * H * 16


@pytest.mark.parametrize('name', NAMES)
def test_image_equals_golden_mode0(require_gpu, name):
    g = np.load(os.path.join(GOLDEN, 'oracle_%s_64x64.npz' % name))
    pt = make_tracer(name, 64, 64, has_sky=int(name != 'cornell-box'), seed=2024)
    pt.render_frame(64)
    got = pt.read_pixels()['color']
    assert np.array_equal(got.view(np.uint32), g['radiance_mode0_64spp'].view(np.uint32))


@pytest.mark.parametrize('name', NAMES)
def test_image_equals_oracle_with_same_streams_mode1(require_gpu, name, flats):
    """Indirect chains (rendering.d:376-394, 474-477): the same rays are traced and the same decisions taken as in the
    oracle (counts equal exactly), and the radiance is the same bits: every chain vertex keeps its own computeLdirect
    sum and the chain is folded from its end, in the order the recursion returns.  cornell-box and chest1 take the
    kernels that trace the rays they spawn, the teapot the queued ones."""
    W = H = 64
    sky = name != 'cornell-box'
    pt = make_tracer(name, W, H, has_sky=int(sky), mode=1, seed=2024)
    o = make_oracle(flats[name], name, W, H, has_sky=sky, mode=1, seed=2024)
    pt.render_frame(16)
    got = pt.read_pixels()
    ref = oracle_py.new_pixels(W * H); _, cn = o.render_keyed(ref, 0, 16, nthreads=8)
    g = np.load(os.path.join(GOLDEN, 'oracle_%s_64x64.npz' % name))
    assert np.allclose(ref['color'], g['radiance_mode1_16spp'], rtol=1e-4, atol=1e-5)
    st = pt.stats()
    rays = st.raysPrimary + st.raysLight + st.raysIndirect
    assert int(rays) == int(cn[0]) and int(st.raysOcclusion) == int(cn[1])
    assert np.array_equal(got['n'], ref['n'])
    assert np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    assert np.array_equal(got['color'].view(np.uint32), g['radiance_mode1_16spp'].view(np.uint32))
    assert np.isfinite(got['sum']).all() and (got['sum'] >= (-0.0036 * 16 * 6 if sky else 0)).all()


@pytest.mark.parametrize('name,sky,depth,chains,rr', [('cornell-box', 1, 2, 3, 0.9), ('cornell-box', 0, 7, 8, 0.8), ('teapot', 1, 1, 2, 0.95),
                                                     ('teapot', 1, 6, 6, 0.5), ('chest1', 1, 3, 1, 0.7)])
def test_chains_of_other_depths_and_counts_equal_the_oracle(require_gpu, name, sky, depth, chains, rr, flats):
    """The records of a chain are laid out by bounce round and chain (kernels.h: Wavefront::Lacc), so their number
    follows maxDepth and numIndirect: depth caps from 1 to 7, one to eight chains, other roulette probabilities --
    same rays as the oracle (counts) and the same bits, on the kernels that trace their own rays (cornell-box, chest1)
    and on the queued ones (teapot)."""
    W, H, spp = 48, 40, 6
    pt = make_tracer(name, W, H, has_sky=sky, mode=1, seed=77, max_depth=depth, n_indirect=chains, rr=rr)
    o = make_oracle(flats[name], name, W, H, has_sky=bool(sky), mode=1, seed=77, max_depth=depth, n_indirect=chains, rr=rr)
    pt.render_frame(spp); pt.render_frame(2)
    got = pt.read_pixels()
    ref = oracle_py.new_pixels(W * H); _, c1 = o.render_keyed(ref, 0, spp, nthreads=8); _, c2 = o.render_keyed(ref, 0, 2, nthreads=8)
    st = pt.stats()
    assert int(st.raysPrimary + st.raysLight + st.raysIndirect) == int(c1[0] + c2[0]) and int(st.raysOcclusion) == int(c1[1] + c2[1])
    assert st.raysIndirect > 0
    assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))


@pytest.mark.parametrize('name,mode,spp', [('cornell-box', 0, 256), ('teapot', 0, 64), ('chest1', 1, 64)])
def test_converged_image_vs_reference_shaped_oracle(require_gpu, name, mode, spp, flats):
    """Equal-spp comparison with the bands + Mt19937 + sphere-tree renderer: within Monte Carlo noise."""
    W = H = 64
    sky = name != 'cornell-box'
    pt = make_tracer(name, W, H, has_sky=int(sky), mode=mode, seed=31)
    pt.render_frame(spp)
    got = pt.read_pixels()['color']
    o = make_oracle(flats[name], name, W, H, build_tree=True, has_sky=sky, mode=mode, seed=1)
    a = oracle_py.new_pixels(W * H); o.render_mt(a, spp, nthreads=8, use_tree=True)
    b = oracle_py.new_pixels(W * H)
    b['n'] = 1000.0                                   # different band seeds (seed = pixelOffset + n, main.d:117)
    o.render_mt(b, spp, nthreads=8, use_tree=True)
    bc = (b['sum'] / spp).astype(np.float32)
    noise = rel_rmse(bc, a['color'])
    assert rel_rmse(got, a['color']) <= 2.0 * noise
    bias = np.abs(box(got, H, W) - box(a['color'], H, W)).mean() / a['color'].mean()
    assert bias <= max(0.01, 1.5 * np.abs(box(bc, H, W) - box(a['color'], H, W)).mean() / a['color'].mean())


def test_batching_progression_and_windows_do_not_change_the_image(require_gpu):
    """Streams are keyed (seed, pixel, sample): any schedule of the same samples gives the same bits."""
    name, W, H, spp = 'cornell-box', 80, 48, 12
    ref = make_tracer(name, W, H, seed=5); ref.render_frame(spp); want = ref.read_pixels()
    small = make_tracer(name, W, H, seed=5, batch_slots=1024)            # many pixel chunks x sample batches
    small.render_frame(spp)
    assert np.array_equal(small.read_pixels()['sum'].view(np.uint32), want['sum'].view(np.uint32))
    prog = make_tracer(name, W, H, seed=5)                                # 5 + 7 passes, as the progressive loop does
    prog.render_frame(5); prog.render_frame(7)
    assert np.array_equal(prog.read_pixels()['sum'].view(np.uint32), want['sum'].view(np.uint32))
    win = make_tracer(name, W, H, seed=5)                                 # computeOutputColor on disjoint slices
    for off in range(0, W * H, 4 * W):                                   # main.d:184-192: 4 scanlines per task
        win.computeOutputColor(off, 4 * W, spp)
    assert np.array_equal(win.read_pixels()['sum'].view(np.uint32), want['sum'].view(np.uint32))
    m1 = make_tracer(name, W, H, seed=5, mode=1, batch_slots=2048); m1.render_frame(4)
    m2 = make_tracer(name, W, H, seed=5, mode=1); m2.render_frame(4)
    assert np.array_equal(m1.read_pixels()['sum'].view(np.uint32), m2.read_pixels()['sum'].view(np.uint32))


def test_full_batches_where_nearly_every_ray_hits(require_gpu, flats):
    """Batches that fill the wavefront's capacity exactly (256x256 pixels = the 65536-slot minimum) in a scene where
    almost every view ray hits: the hit queue receives as many entries as the ray queue held, which is the case the
    per-workgroup output regions of extend are sized for.  Bit-identical to the oracle with the same streams."""
    name, W, H, spp = 'cornell-box', 256, 256, 3
    pt = make_tracer(name, W, H, seed=77, batch_slots=65536)
    pt.render_frame(spp)
    got = pt.read_pixels()
    o = make_oracle(flats[name], name, W, H, has_sky=False, seed=77)
    ref = oracle_py.new_pixels(W * H); o.render_keyed(ref, 0, spp, nthreads=8)
    assert (got['n'] == spp).all()
    assert np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    hit = pt.trace_primary()[0] >= 0
    assert hit.mean() > 0.95


def test_background_with_the_sun_disc_in_view_is_bit_identical(require_gpu):
    """Camera looking along sampleSky's sun direction (rendering.d:332-339): background pixels inside or at the edge
    of the disc are evaluated per sample, those away from it are filled by the constant-sky shortcut; both must equal
    the oracle bit for bit (no sin/cos is involved in a miss).  Also with the masks switched off."""
    from conftest import soup, UP
    W, H, spp = 96, 64, 8
    fl = soup(300, seed=4, extent=6.0)
    cam = oracle_py.camera_lookat(30.0, H / W, (-20, -20, -120), (0, 0, 0), UP)
    o = oracle_py.Oracle(fl, build_tree=False)
    o.configure(width=W, height=H, has_sky=True, seed=11)
    o.set_camera(cam)
    ref = oracle_py.new_pixels(W * H); o.render_keyed(ref, 0, spp, nthreads=8)
    for flags in (0, P.PT_FLAG_NO_COVERAGE_MASK):
        pt = P.PathTracer(W, H, device=0, has_sky=1, seed=11, flags=flags)
        pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
        pt.set_camera(cam)
        covered, const_sky = pt.coverage_mask(constant_sky=True)
        assert const_sky.any() and (~covered & ~const_sky).any()          # both kinds of background pixel occur
        pt.render_frame(spp)
        got = pt.read_pixels()
        bg = ~covered
        assert (got['n'] == spp).all()
        assert np.array_equal(got['sum'][bg].view(np.uint32), ref['sum'][bg].view(np.uint32))
        assert (got['sum'][bg].max(1) > 30 * spp * 0.5).any()              # the disc itself (radiance 30) is in there
        assert np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
        st = pt.stats()
        assert st.raysPrimary == W * H * spp and st.raysCulled >= (bg.sum() * spp if flags == 0 else 0)


def test_scene_without_hittable_triangles_renders_the_sky(require_gpu):
    """Zero triangles, and triangles that can never be hit (degenerate: |det| < FloatEpsilon for every ray): every
    sample takes computeL's else branch (rendering.d:480)."""
    from conftest import UP
    W, H, spp = 64, 40, 5
    mats = np.array([[0.0, 0.7, 0.7, 0.7]], np.float32)
    line = np.array([[[0, 0, 0], [1, 1, 1], [2, 2, 2]]], np.float32)          # collinear vertices
    for v in (np.zeros((0, 3, 3), np.float32), line):
        for flags in (0, P.PT_FLAG_NO_COVERAGE_MASK):
            pt = P.PathTracer(W, H, device=0, has_sky=1, seed=1, flags=flags)
            pt.upload_scene(v, np.zeros_like(v), np.zeros(len(v), np.uint32), mats)
            pt.set_camera(oracle_py.camera_lookat(30.0, H / W, (5, 4, -3), (0, 0, 0), UP))    # looks down: no sun disc
            pt.render_frame(spp)
            px = pt.read_pixels()
            assert (px['n'] == spp).all()
            assert np.allclose(px['color'], np.array([0.682, 0.977, 1.0], np.float32), rtol=1e-6)
            st = pt.stats()
            assert st.raysOcclusion == 0 and st.raysPrimary == W * H * spp


def test_band_and_sample_splits_reassemble_bit_identically(require_gpu):
    """Two handles stand in for two GPUs: interleaved 4-scanline bands summed == one GPU; sample slices summed ==
    the per-slice images added."""
    name, W, H, spp = 'teapot', 96, 52, 8
    one = make_tracer(name, W, H, has_sky=1, seed=8); one.render_frame(spp); want = one.read_pixels()
    parts = []
    for r in range(3):
        t = make_tracer(name, W, H, has_sky=1, seed=8)
        t.render_bands(4, r, 3, spp)
        parts.append(t.read_pixels())
    n = sum(p['n'] for p in parts); s = sum(p['sum'] for p in parts)
    assert (n == spp).all() and np.array_equal(s.view(np.uint32), want['sum'].view(np.uint32))
    rows = parts[1]['n'].reshape(H, W)
    assert (rows[4:8] == spp).all() and (rows[0:4] == 0).all() and (rows[8:12] == 0).all()
    # sample slices: rank r of G renders sample indices r, r + G, r + 2G, ... (sampleBase = r, sampleStride = G), in
    # two passes with its own accumulator, as a progressive multi-GPU render does; the index sets of the ranks are
    # disjoint and together they are the 2 * G * spp indices one renderer draws with stride 1
    G = 2
    single = []                                                           # radiance of each sample index alone
    for k in range(2 * G * 2):
        t = make_tracer(name, W, H, has_sky=1, seed=8, sample_base=k); t.render_frame(1); single.append(t.read_pixels()['sum'])
    for r in range(G):
        t = make_tracer(name, W, H, has_sky=1, seed=8, sample_base=r, sample_stride=G)
        t.render_frame(2); t.render_frame(2)
        acc = np.zeros_like(single[0])
        for k in range(r, 2 * G * 2, G):
            acc = acc + single[k]
        got = t.read_pixels()
        assert (got['n'] == 4).all() and np.array_equal(got['sum'].view(np.uint32), acc.view(np.uint32))


@pytest.mark.parametrize('mode', [0, 1])
def test_tile_strips_of_a_scene_in_global_memory_do_not_change_the_image(require_gpu, mode):
    """A scene traversed from global memory numbers its pixel tiles by vertical strips of 32 tiles (frames wider than
    256 pixels), and its shadow kernels hand lanes the next ray.  Whole frames, windows that are not whole rows, bands
    of a frame whose width is not a multiple of the strip (a narrower last strip) and small batches (several chunks per
    frame): the same bits as the same scene rendered from LDS, where tiles are numbered row by row.  By default the hits
    of such a scene are shaded in the order of the grid cells they lie in (view-ray hits: a few cells behind a few tiles;
    hits of the chains in mode 1: cells all over the scene, which the kernel coarsens until they are at most 256);
    PT_FLAG_NO_HIT_ORDER keeps the order of the rays: the same bits again."""
    fl = soup(1200, seed=31, size=14.0)
    W, H, spp = 808, 96, 3                                                # 101 tiles per row: three strips of 32 and one of 5
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
    out = []
    G = P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_HIT_ORDER                     # (the order is the default only for finer geometry than this)
    for flags, slots in ((0, 0), (G, 0), (G, 65536), (G | P.PT_FLAG_NO_HIT_ORDER, 0), (G | P.PT_FLAG_NO_HIT_ORDER, 65536)):
        pt = P.PathTracer(W, H, has_sky=1, mode=mode, seed=12, max_depth=2, flags=flags, batch_slots=slots)
        pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials']); pt.set_camera(cam)
        assert pt.stats().sceneInLds == (0 if flags else 1) and pt.stats().numLightGroups >= 1
        pt.render_frame(spp)
        pt.render(W * 5 + 3, W * 40 + 11, 2)
        pt.render_bands(4, 1, 2, 2)
        pt.render_bands(8, 0, 3, 1)
        out.append((pt.read_pixels(), pt.stats()))
        pt.close()
    (a, sa) = out[0]
    assert a['n'].max() == spp + 2 + 2 + 1
    for x, sx in out[1:]:
        assert np.array_equal(a['n'], x['n']) and np.array_equal(a['sum'].view(np.uint32), x['sum'].view(np.uint32))
        for f in ('raysPrimary', 'raysLight', 'raysIndirect', 'raysOcclusion', 'hits', 'hitsPrimary', 'envEntries', 'samples'):
            assert getattr(sa, f) == getattr(sx, f), f


def test_hit_order_is_the_default_only_under_fine_geometry(require_gpu):
    """A scene in global memory with at least half as many triangles as the frame has pixels gets its hits put in the order
    of their cells without being asked (one more launch per stage than with PT_FLAG_NO_HIT_ORDER); under coarser geometry
    only PT_FLAG_HIT_ORDER does that.  The image is the same bits every way."""
    fl = soup(1200, seed=8, size=14.0)
    k = list(P.PT_K_NAMES).index('other')
    for W, H, auto in ((48, 40, True), (160, 96, False)):                    # 2400 >= 1920 pixels; 2400 < 15360
        cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
        out = []
        for extra in (0, P.PT_FLAG_HIT_ORDER, P.PT_FLAG_NO_HIT_ORDER):
            pt = P.PathTracer(W, H, has_sky=1, mode=0, seed=4, flags=P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_TIME_KERNELS | extra)
            pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials']); pt.set_camera(cam)
            pt.render_frame(3)
            out.append((pt.read_pixels()['sum'].view(np.uint32).copy(), pt.stats().kernelLaunches[k]))
            pt.close()
        (a, na), (b, nb), (c, nc) = out
        assert np.array_equal(a, b) and np.array_equal(a, c)
        assert nb == nc + 1 and na == (nb if auto else nc), (na, nb, nc)


def test_pixel_checkpoint_roundtrip_and_resume(require_gpu, tmp_path):
    """main.d:227-234, 333-340: raw Pixel[] dump and restore; resuming continues the same sample sequence."""
    name, W, H = 'cornell-box', 64, 40
    full = make_tracer(name, W, H, seed=3); full.render_frame(10); want = full.read_pixels()
    a = make_tracer(name, W, H, seed=3); a.render_frame(4)
    path = tmp_path / 'sponza.project'
    a.read_pixels().tofile(path)
    assert os.path.getsize(path) == W * H * 28
    b = make_tracer(name, W, H, seed=3)
    b.write_pixels(np.fromfile(path, dtype=P.PIXEL_DTYPE))
    b.render_frame(6)
    got = b.read_pixels()
    assert np.array_equal(got['sum'].view(np.uint32), want['sum'].view(np.uint32)) and (got['n'] == 10).all()
    assert np.allclose(got['color'], got['sum'] / got['n'][:, None], rtol=1e-7)
    b.clear_pixels()
    z = b.read_pixels()
    assert (z['n'] == 0).all() and (z['sum'] == 0).all() and (z['color'] == 0).all()


def test_tonemap_equals_oracle(require_gpu):
    name, W, H = 'teapot', 64, 48
    pt = make_tracer(name, W, H, has_sky=1, seed=1); pt.render_frame(8)
    px = pt.read_pixels()
    rgb = pt.tonemap()
    ref = oracle_py.tonemap(px.view(oracle_py.PIXEL_DTYPE))
    assert np.array_equal(rgb, ref)                                      # bytes: bit-exact
    assert rgb.max() == 254                                              # saturated sky: 1.055f - 0.055f truncates to 254


def test_full_size_properties(require_gpu):
    """Size-independent checks at the benchmark frame size (1920x1080), 4 spp: every pixel gets exactly its samples,
    nothing negative or non-finite, misses carry the sky constant, ray counts add up."""
    name, W, H, spp = 'teapot', 1920, 1080, 4
    pt = make_tracer(name, W, H, has_sky=1, seed=0x5EED)
    tri = pt.trace_primary()[0]
    pt.reset_stats()
    pt.render_frame(spp)
    px = pt.read_pixels()
    assert (px['n'] == spp).all() and np.isfinite(px['sum']).all()
    # the sun term multiplies by n.g_sunDir although the visibility test used n.sampleDir (rendering.d:436,442): on
    # surfaces within 0.9 degrees of tangent to the sun it is slightly negative, by at most sin(0.9 deg)*0.7/pi a sample
    assert (px['sum'] >= -0.0036 * spp).all()
    st = pt.stats()
    assert st.samples == W * H * spp and st.raysPrimary == W * H * spp and st.raysLight == 0 and st.raysIndirect == 0
    assert 0 < st.raysOcclusion <= 2 * (tri >= 0).sum() * spp * 1.05
    interior_miss = np.ones(W * H, bool)
    hit2d = (tri >= 0).reshape(H, W)
    grown = hit2d.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            grown |= np.roll(np.roll(hit2d, dy, 0), dx, 1)
    interior_miss = ~grown.reshape(-1)
    sky = np.array([0.682, 0.977, 1.0], np.float32)                      # sampleSky away from its sun disc, rendering.d:338
    c = px['color'][interior_miss]
    is_sky = np.all(np.abs(c - sky) < 1e-6, axis=1)
    is_sun = np.all(c > 0.99, axis=1)
    assert (is_sky | is_sun | (c.max(1) > 1)).all() and is_sky.mean() > 0.99


@pytest.mark.parametrize('flags', [0, P.PT_FLAG_NO_LDS_SCENE])
@pytest.mark.parametrize('mode', [0, 1])
def test_lights_and_sky_together_equal_oracle_with_same_streams(require_gpu, flags, mode):
    """A scene with a light group AND the sky: the light-sample kernel adds to a vertex's radiance first, the sun/sky
    kernel then reads, adds and writes it back (it may store it outright only when nothing precedes it).  Same streams
    as the oracle: bit-identical in both modes (mode 1 through the queued kernels, scene in LDS and in global memory)."""
    fl = soup(1500, seed=21, size=12.0)                   # big triangles: most view rays hit, many shadow rays are blocked
    W, H = 96, 64
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
    pt = P.PathTracer(W, H, has_sky=1, mode=mode, seed=5, flags=flags)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials']); pt.set_camera(cam)
    assert pt.stats().numLightGroups >= 1 and pt.stats().sceneInLds == (0 if flags else 1)
    o = oracle_py.Oracle(fl, build_tree=False); o.configure(width=W, height=H, has_sky=True, mode=mode, seed=5); o.set_camera(cam.state)
    spp = 8
    pt.render_frame(spp); got = pt.read_pixels()
    ref = oracle_py.new_pixels(W * H); _, cn = o.render_keyed(ref, 0, spp, nthreads=8)
    assert np.array_equal(got['n'], ref['n'])
    st = pt.stats()
    assert st.raysLight > 0 and st.raysOcclusion > 0
    assert int(st.raysPrimary + st.raysLight + st.raysIndirect) == int(cn[0]) and int(st.raysOcclusion) == int(cn[1])
    assert np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))


def test_frame_transfers_pinned_staged_and_overlapped(require_gpu):
    """Pixel[] read-back / restore and the tone-mapped frame give the same bytes whether the caller's buffer is
    page-locked (copied into directly) or pageable (through the handle's staging buffer), and whether the tone map is
    fetched in one call or in two halves with the next pass rendered in between (main.d:273-287's loop shape)."""
    name, W, H = 'teapot', 200, 120
    pt = make_tracer(name, W, H, has_sky=1, seed=4); pt.render_frame(3)
    plain = pt.read_pixels()
    pinned = pt.host_array(W * H, P.PIXEL_DTYPE)
    pt._check(pt.lib.ptgpu_read_pixels(pt.h, 0, W * H, pinned.ctypes.data))
    assert np.array_equal(plain.view(np.uint8), pinned.view(np.uint8))
    part = pt.read_pixels(1000, 777)
    assert np.array_equal(part.view(np.uint8), plain[1000:1777].view(np.uint8))
    rgb = pt.tonemap()
    a = np.zeros((W * H, 3), np.uint8); b = pt.host_array((W * H, 3), np.uint8)
    other = make_tracer(name, W, H, has_sky=1, seed=4); other.render_frame(3)
    for buf in (a, b):                                              # pageable, then page-locked destination
        pt.tonemap_begin(buf)
        pt.render_frame(1)                                          # the next pass, queued while the copy is in flight
        pt.tonemap_end()
        assert np.array_equal(buf, rgb)                             # the frame as it was when begin was called
        other.render_frame(1)
        rgb = pt.tonemap()
        assert np.array_equal(rgb, other.tonemap())
    with pytest.raises(P.PtError):
        pt.tonemap_end()                                            # nothing outstanding
    pt.tonemap_begin(a)
    with pytest.raises(P.PtError):
        pt.tonemap_begin(b)                                         # one at a time
    pt.tonemap_end()
    # restore from a page-locked and from a pageable array
    fresh = make_tracer(name, W, H, has_sky=1, seed=4)
    fresh.write_pixels(plain); one = fresh.read_pixels()
    fresh.clear_pixels()
    fresh._check(fresh.lib.ptgpu_write_pixels(fresh.h, 0, W * H, pinned.ctypes.data)); two = fresh.read_pixels()
    assert np.array_equal(one.view(np.uint8), plain.view(np.uint8)) and np.array_equal(two.view(np.uint8), plain.view(np.uint8))
    pt.host_free(pinned); pt.host_free(b)


@pytest.mark.parametrize('split', ['bands', 'samples'])
def test_native_group_of_one_gpu_equals_a_single_handle(require_gpu, split):
    """ptgpu_group_* with one GPU (all a one-GPU box can hold): share-out, two passes, the exchange (RCCL with a
    communicator of one rank) and the combined frame, which equals a plain handle's render bit for bit."""
    name, W, H, spp = 'teapot', 160, 90, 3
    g = P.GpuGroup(W, H, [0], has_sky=1, seed=21)
    assert g.info() == {'devices': 1, 'root': 0, 'rccl': True}
    g.Scene(conftest_model(name))
    cam = P.Camera(30.0, H / W).setTransform(CAMERAS[name][0], CAMERAS[name][1], (0, 0, 1))
    g.set_camera(cam)
    g.render(spp, split=split); g.render(2, split=split)
    frame = g.combine()
    got = frame.read_pixels()
    one = make_tracer(name, W, H, has_sky=1, seed=21); one.render_frame(spp + 2)
    want = one.read_pixels()
    assert np.array_equal(got['n'], want['n']) and np.array_equal(got['sum'].view(np.uint32), want['sum'].view(np.uint32))
    assert np.array_equal(frame.tonemap(), one.tonemap())
    with pytest.raises(P.PtError):
        g.render(1, split='samples' if split == 'bands' else 'bands')      # the split is fixed until clear
    g.clear()
    g.render(1, split='samples' if split == 'bands' else 'bands')
    assert (g.combine().read_pixels()['n'] == 1).all()
    g.close()


def conftest_model(name):
    from conftest import model_path
    return model_path(name)


@pytest.mark.parametrize('name,sky,mode', [('cornell-box', 0, 0), ('teapot', 1, 0), ('chest1', 1, 1), ('cornell-box', 0, 1)])
def test_repeated_small_passes_replayed_from_a_graph_equal_one_render(require_gpu, name, sky, mode):
    """The progressive loop (main.d:273-287) issues the same one-sample pass again and again; from its second
    occurrence such a call is captured into a HIP graph and replayed.  Six passes equal one render of six samples bit for
    bit, also with the graph switched off, with a window in between (another shape: no replay) and after a camera change."""
    W, H = 160, 120
    want = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17, flags=P.PT_FLAG_NO_GRAPH); want.render_frame(6)
    ref = want.read_pixels()
    pt = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17)
    for _ in range(3):
        pt.render_frame(1)
    pt.render(0, 7 * W, 1)                                          # a different shape in between
    for _ in range(3):
        pt.render_frame(1)
    pt.render(7 * W, (H - 7) * W, 1)
    got = pt.read_pixels()
    assert (got['n'] == 7).all()
    full = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17, flags=P.PT_FLAG_NO_GRAPH); full.render_frame(7)
    assert np.array_equal(got['sum'].view(np.uint32), full.read_pixels()['sum'].view(np.uint32))
    six = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17)
    for _ in range(6):
        six.render_frame(1)
    assert np.array_equal(six.read_pixels()['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    # camera change: the captured launches hold the old camera, so the shape is captured anew
    cam2 = P.Camera(30.0, H / W).setTransform((CAMERAS[name][0][0] + 1.0, CAMERAS[name][0][1], CAMERAS[name][0][2]), CAMERAS[name][1], (0, 0, 1))
    a = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17); b = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=17, flags=P.PT_FLAG_NO_GRAPH)
    for t in (a, b):
        for _ in range(3):
            t.render_frame(1)
        t.clear_pixels(); t.set_camera(cam2)
        for _ in range(3):
            t.render_frame(1)
    assert np.array_equal(a.read_pixels()['sum'].view(np.uint32), b.read_pixels()['sum'].view(np.uint32))


@pytest.mark.parametrize('name,sky,mode', [('cornell-box', 0, 0), ('chest1', 1, 1)])
def test_repeated_small_passes_on_a_stream_of_the_caller(require_gpu, name, sky, mode):
    """ptgpu_set_stream hands the handle a stream of the caller's -- here the legacy null stream (what
    torch.cuda.current_stream() is by default), which cannot be captured, and a torch stream.  Repeated identical small
    passes run as ordinary launches there; three passes equal one render of three samples."""
    import torch
    W, H = 160, 120
    want = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=23, flags=P.PT_FLAG_NO_GRAPH); want.render_frame(4)
    ref = want.read_pixels()
    side = torch.cuda.Stream()
    for stream in (0, side.cuda_stream):
        pt = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=23)
        pt.set_stream(stream)
        for _ in range(4):
            pt.render_frame(1)
        got = pt.read_pixels()
        assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
        pt.close()


@pytest.mark.parametrize('name,sky,mode', [('chest1', 1, 0), ('chest1', 1, 1), ('cornell-box', 0, 1), ('cornell-box', 1, 1),
                                           ('cornell-box', 1, 0)])
def test_shading_that_traces_its_own_rays_equals_the_queued_path(require_gpu, name, sky, mode):
    """Small scenes: the shading kernel traces the rays it spawns (light-sample rays, sun and sky rays, continuation
    rays) instead of queueing them for connect / occluded / extend.  Same rays, same arithmetic in the same order: the
    accumulators are the same bits as with PT_FLAG_NO_FUSE, every counter is equal, and the counters say which path ran."""
    W, H, spp = 96, 64, 12
    out = {}
    for flags in (0, P.PT_FLAG_NO_FUSE, P.PT_FLAG_QUEUED_VIEW_RAYS):
        pt = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=77, max_depth=4, flags=flags | P.PT_FLAG_TIME_KERNELS)
        pt.render_frame(spp)
        pt.render_frame(3)                                # a second batch: chain records of the first carry an older tag
        st = pt.stats()
        out[flags] = (pt.read_pixels(), st)
        pt.close()
    (a, sa), (b, sb), (c, sc) = out[0], out[P.PT_FLAG_NO_FUSE], out[P.PT_FLAG_QUEUED_VIEW_RAYS]
    for x, sx in ((b, sb), (c, sc)):
        assert np.array_equal(a['n'], x['n']) and np.array_equal(a['sum'].view(np.uint32), x['sum'].view(np.uint32))
        for f in ('raysPrimary', 'raysLight', 'raysIndirect', 'raysOcclusion', 'raysCulled', 'hits', 'hitsPrimary', 'envEntries', 'samples'):
            assert getattr(sa, f) == getattr(sx, f), f
    assert sa.raysPrimaryInShade == sa.raysPrimary - sa.raysCulled > 0 and sb.raysPrimaryInShade == 0 and sc.raysPrimaryInShade == 0
    assert sc.raysIndirectInShade == sa.raysIndirectInShade and sc.raysOcclusionInShade == sa.raysOcclusionInShade
    assert sb.raysIndirectInShade == 0 and sb.raysOcclusionInShade == 0 and sb.raysLightInShade == 0
    lights = name == 'cornell-box'
    assert sa.raysIndirectInShade == sa.raysIndirect and (sa.raysIndirect > 0) == (mode == 1)
    assert sa.raysOcclusionInShade == sa.raysOcclusion and (sa.raysOcclusion > 0) == bool(sky)
    assert sa.raysLightInShade == sa.raysLight and (sa.raysLight > 0) == lights
    k = list(P.PT_K_NAMES)
    assert sa.kernelLaunches[k.index('extend')] == 0 and sc.kernelLaunches[k.index('extend')] == 2      # view rays of two batches
    assert sb.kernelLaunches[k.index('extend')] == (2 if mode == 0 else 2 * (2 + 4))
    assert sa.kernelLaunches[k.index('occluded')] == 0 and sa.kernelLaunches[k.index('connect')] == 0
    assert sa.kernelLaunches[k.index('sample_traced')] == 2 and sa.kernelLaunches[k.index('shade_traced')] == (2 if mode == 1 else 0)
    assert sa.raysIndirectInSample + sa.raysOcclusionInSample + sa.raysLightInSample > 0 and sc.raysIndirectInSample == 0
    assert sa.raysIndirectInSample <= sa.raysIndirectInShade and sa.raysOcclusionInSample <= sa.raysOcclusionInShade
    assert (sb.kernelLaunches[k.index('occluded')] > 0) == bool(sky) and (sb.kernelLaunches[k.index('connect')] > 0) == lights


@pytest.mark.parametrize('name,sky,mode,W,H,spp', [('chest1', 1, 1, 512, 384, 10), ('cornell-box', 1, 1, 512, 384, 10),
                                                   ('cornell-box', 0, 0, 512, 384, 10), ('chest1', 1, 1, 1024, 768, 44),
                                                   ('cornell-box', 1, 1, 1024, 768, 44)])
def test_samples_taken_from_the_common_pool_equal_the_workgroups_own_shares(require_gpu, name, sky, mode, W, H, spp):
    """The one-kernel-per-sample path on frames large enough for the pool of 64-sample runs that whichever workgroup
    is free takes from (the smaller frame pools a quarter of its runs, the larger one all of them): same bits and
    counters as with every workgroup keeping to its own share (PT_FLAG_STATIC_RUNS) and as the queued path, over two
    batches (the pool's counters are cleared in between)."""
    out = []
    for flags in (0, P.PT_FLAG_STATIC_RUNS, P.PT_FLAG_NO_FUSE):
        pt = make_tracer(name, W, H, has_sky=sky, mode=mode, seed=5, max_depth=3, flags=flags)
        pt.render_frame(spp)
        pt.render_frame(7)
        out.append((pt.read_pixels(), pt.stats()))
        pt.close()
    (a, sa) = out[0]
    assert sa.raysPrimaryInShade > 0
    for x, sx in out[1:]:
        assert np.array_equal(a['n'], x['n']) and np.array_equal(a['sum'].view(np.uint32), x['sum'].view(np.uint32))
        for f in ('raysPrimary', 'raysLight', 'raysIndirect', 'raysOcclusion', 'raysCulled', 'hits', 'hitsPrimary', 'envEntries', 'samples'):
            assert getattr(sa, f) == getattr(sx, f), f


@pytest.mark.parametrize('name,sky,W,H,spp', [('cornell-box', 0, 1280, 960, 9), ('cornell-box', 1, 1280, 960, 5), ('chest1', 1, 1280, 720, 7),
                                              ('chest1', 1, 320, 200, 1), ('cornell-box', 0, 301, 77, 1)])
def test_samples_added_up_in_registers_equal_the_recorded_ones(require_gpu, name, sky, W, H, spp):
    """Mode 0 on the one-kernel-per-sample path, windows of at least 786432 pixels or one sample per pixel: a lane
    renders the samples of one pixel one after the other and adds them to the pixel's sum in registers
    (k_pixel_traced) -- no radiance record per sample, no accumulate launch.  Same bits and counters as the recorded
    path (PT_FLAG_NO_PIXEL_SUMS) and as one launch per stage (PT_FLAG_NO_FUSE), over two calls (the second continues the
    pixels' sample numbering), for whole frames, windows that are not whole tiles, and interleaved bands."""
    out = []
    for flags in (0, P.PT_FLAG_NO_PIXEL_SUMS, P.PT_FLAG_NO_FUSE, P.PT_FLAG_STATIC_RUNS):
        pt = make_tracer(name, W, H, has_sky=sky, mode=0, seed=5, flags=flags | P.PT_FLAG_TIME_KERNELS)
        pt.render_frame(spp)
        pt.render_frame(1)
        pt.render(W * 3 + 5, W * H - W * 7 - 11, spp)               # a window that starts and ends inside a row
        pt.render_bands(4, 1, 3, spp)
        out.append((pt.read_pixels(), pt.stats()))
        pt.close()
    (a, sa) = out[0]
    k = dict(zip(P.PT_K_NAMES, sa.kernelLaunches))
    # the two frames and (large frames, or one sample per pixel) the window and the bands; a third of a frame with
    # several samples per pixel is rendered the recorded way
    assert k['pixel_traced'] >= (3 if W * H >= 786432 or spp == 1 else 2) and sa.raysPrimaryInShade > 0
    kb = dict(zip(P.PT_K_NAMES, out[1][1].kernelLaunches))
    assert kb['pixel_traced'] == 0 and kb['sample_traced'] > 0 and kb['accumulate'] > 0
    for x, sx in out[1:]:
        assert np.array_equal(a['n'], x['n']) and np.array_equal(a['sum'].view(np.uint32), x['sum'].view(np.uint32))
        for f in ('raysPrimary', 'raysLight', 'raysIndirect', 'raysOcclusion', 'raysCulled', 'hits', 'hitsPrimary', 'envEntries', 'samples',
                  'samplesConstantSky'):
            assert getattr(sa, f) == getattr(sx, f), f


def test_wavefront_state_follows_the_path_the_scene_takes(require_gpu):
    """A handle whose scene takes the one-kernel-per-sample path keeps neither the view-ray queue nor the shadow queues.
    The same handle switched to the queued path (ptgpu_set_config), back again, and given a scene too large for that
    path (the teapot) renders the same bits as fresh handles do."""
    from conftest import CAMERAS, UP, model_path
    W, H, spp = 128, 96, 5

    def fresh(name, flags):
        t = make_tracer(name, W, H, has_sky=1, mode=1, seed=9, max_depth=3, flags=flags)
        t.render_frame(spp)
        px = t.read_pixels(); t.close()
        return px['sum'].view(np.uint32)

    want_small, want_big = fresh('chest1', 0), fresh('teapot', 0)
    assert np.array_equal(want_small, fresh('chest1', P.PT_FLAG_NO_FUSE))
    pt = make_tracer('chest1', W, H, has_sky=1, mode=1, seed=9, max_depth=3)
    for flags in (0, P.PT_FLAG_NO_FUSE, 0, P.PT_FLAG_QUEUED_VIEW_RAYS, 0):
        pt.set_config(flags=flags)
        pt.clear_pixels(); pt.render_frame(spp)
        assert np.array_equal(pt.read_pixels()['sum'].view(np.uint32), want_small), flags
    pt.Scene(model_path('teapot'))
    pt.set_camera(P.Camera(30.0, H / W).setTransform(CAMERAS['teapot'][0], CAMERAS['teapot'][1], UP))
    pt.clear_pixels(); pt.render_frame(spp)
    assert np.array_equal(pt.read_pixels()['sum'].view(np.uint32), want_big)
    pt.close()


@pytest.mark.parametrize('flags', [0, P.PT_FLAG_NO_LDS_SCENE])
def test_traversal_kernel_that_shades_its_hits_equals_the_queued_path(require_gpu, flags):
    """Mode 0, sky, no light group, a scene too large for the shading kernel's LDS (the teapot): the kernel that traces
    the sun and sky rays shades the view-ray hits itself, from the hit queue.  Same bits and counters as shade followed
    by occluded (PT_FLAG_NO_FUSE), and no shade launch.  Traversed from global memory the scene keeps shade and the
    sun / sky kernel apart (that kernel's lanes take the next ray there): the same bits once more."""
    W, H, spp = 160, 96, 6
    out = []
    for extra in (0, P.PT_FLAG_NO_FUSE):
        pt = make_tracer('teapot', W, H, has_sky=1, mode=0, seed=3, flags=flags | extra | P.PT_FLAG_TIME_KERNELS)
        pt.render_frame(spp); pt.render_frame(2)
        out.append((pt.read_pixels(), pt.stats()))
        pt.close()
    (a, sa), (b, sb) = out
    assert np.array_equal(a['n'], b['n']) and np.array_equal(a['sum'].view(np.uint32), b['sum'].view(np.uint32))
    for f in ('raysPrimary', 'raysOcclusion', 'raysCulled', 'hits', 'hitsPrimary', 'envEntries', 'samples'):
        assert getattr(sa, f) == getattr(sb, f), f
    assert sa.raysOcclusion > 0 and sa.hits > 0
    k = list(P.PT_K_NAMES)
    fused = 0 if flags & P.PT_FLAG_NO_LDS_SCENE else 2
    assert sa.kernelLaunches[k.index('shade')] == 2 - fused and sb.kernelLaunches[k.index('shade')] == 2
    assert sa.kernelLaunches[k.index('shade_occluded')] == fused and sa.kernelLaunches[k.index('occluded')] == 2 - fused
    assert sb.kernelLaunches[k.index('occluded')] == 2 and sb.kernelLaunches[k.index('shade_occluded')] == 0
