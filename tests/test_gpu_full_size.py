"""The configurations of BASELINE.json at (or near) their full sizes, through the C ABI, checked by what does not
depend on the size: sampled hits against the exhaustive oracle, windows of the frame against the oracle's image,
sample counts, finiteness, ray accounting.  And the native caller: the headless driver's image and project file."""
import os

import numpy as np
import pytest

import conftest
import oracle_py
import pathtracer_amd as P
from conftest import CAMERAS, UP, make_oracle, make_tracer

pytestmark = pytest.mark.gpu


def test_c5_soup_of_ten_million_triangles_at_3840x2160(require_gpu):
    """configs[4]: the 10 M-triangle soup, default flags -- so the BVH is built on the GPU (Morton-order LBVH), the
    scene is traversed from global memory through the 16-byte quantised nodes, and triangle references need more than
    23 bits.  Primary hits of sampled pixel runs equal the exhaustive oracle bit for bit; one sample per pixel of the
    whole 3840x2160 frame adds up."""
    from pathtracer_amd.synthetic import soup
    T = 10_000_000
    fl = soup(T)
    W, H = 3840, 2160
    pt = P.PathTracer(W, H, device=0, has_sky=1, seed=0x5EED)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), UP)
    pt.set_camera(cam)
    st = pt.stats()
    assert st.numTriangles == T + 2 and st.bvhOnGpu == 1 and st.sceneInLds == 0 and st.numLightGroups == 1
    assert st.numNodes > T and st.bvhDepth <= 60
    tri, t, u, v = pt.trace_primary()
    hit = tri >= 0
    assert 0.5 < hit.mean() < 0.8 and tri.max() > (1 << 23)
    # 30 runs of 8 consecutive pixels (the oracle tests every triangle for each of them, 8 threads)
    o = oracle_py.Oracle(fl, build_tree=False)
    o.configure(width=W, height=H, has_sky=True, seed=0x5EED)
    o.set_camera(cam.state)
    rng = np.random.default_rng(3)
    rows = rng.integers(0, H, 30); cols = rng.integers(0, W - 8, 30)
    rows[:4] = [0, H - 1, H // 2, H // 2]; cols[:4] = [0, W - 8, W // 2, 0]
    nhit = 0
    for r, c in zip(rows, cols):
        off = int(r) * W + int(c)
        otri, ot, ou, ov = o.trace_primary(off, 8, use_tree=False)
        assert np.array_equal(tri[off:off + 8], otri)
        h8 = otri >= 0
        nhit += int(h8.sum())
        for got, ref in ((t, ot), (u, ou), (v, ov)):
            assert np.array_equal(got[off:off + 8][h8].view(np.uint32), ref[h8].view(np.uint32))
    assert nhit > 60
    pt.reset_stats()
    pt.render_frame(1)
    px = pt.read_pixels()
    assert (px['n'] == 1).all() and np.isfinite(px['sum']).all() and (px['sum'] >= -0.0036).all()
    st = pt.stats()
    assert st.samples == W * H and st.raysPrimary == W * H and st.raysIndirect == 0
    assert 0 < st.raysLight <= hit.sum() * 1.02 and 0 < st.raysOcclusion <= 2 * hit.sum() * 1.02
    sky = np.array([0.682, 0.977, 1.0], np.float32)
    far_miss = ~hit
    c = px['color'][far_miss]
    assert (np.all(np.abs(c - sky) < 1e-6, axis=1) | (c.min(1) > 0.99)).mean() > 0.97     # misses: sky, or its sun disc (jitter may hit near edges)


def test_c3_chest1_1080p_mode1(require_gpu, flats):
    """configs[2]: chest1 at 1920x1080 with the indirect chains (mode 1), a few samples per pixel.  A band of the frame
    equals the oracle's image of that band bit for bit; the frame adds up."""
    name, W, H, spp = 'chest1', 1920, 1080, 3
    pt = make_tracer(name, W, H, has_sky=1, mode=1, seed=0x5EED)
    pt.render_frame(spp)
    px = pt.read_pixels()
    st = pt.stats()
    assert (px['n'] == spp).all() and np.isfinite(px['sum']).all() and (px['sum'] >= -0.0036 * spp * 6).all()
    tri = pt.trace_primary()[0]
    hits = int((tri >= 0).sum())
    assert st.samples == W * H * spp and st.raysPrimary == W * H * spp and st.raysLight == 0
    # five chains per view-ray hit; a chain shoots its first ray with probability 0.7 and goes on, with probability 0.7
    # each time and up to depth 5, only while its rays hit something (rendering.d:378-393): between the first-ray
    # count and the count of a closed scene, 5 * sum 0.7^k, k = 1..6
    first = hits * spp * 5 * 0.7
    assert 0.98 * first < st.raysIndirect < 1.02 * first * sum(0.7 ** k for k in range(0, 6))
    band0, rows = 500, 24
    o = make_oracle(flats[name], name, W, H, has_sky=True, mode=1, seed=0x5EED)
    ref = oracle_py.new_pixels(rows * W)
    o.render_keyed(ref, band0 * W, spp, nthreads=8)
    got = px[band0 * W:(band0 + rows) * W]
    assert (tri[band0 * W:(band0 + rows) * W] >= 0).mean() > 0.2
    assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))


def test_c4_cornell_3840x2160_and_its_band_split(require_gpu, flats):
    """configs[3]: the cornell box at 3840x2160 (2 of its 4096 spp).  The frame adds up; a 16-row band equals the
    oracle's image of that band bit for bit; and the same frame rendered as the reference's 4-scanline tasks
    (main.d:184-192) shared out over two handles -- the per-GPU shares of a two-GPU run -- reassembles to the same bits."""
    name, W, H, spp = 'cornell-box', 3840, 2160, 2
    pt = make_tracer(name, W, H, seed=0x5EED)
    pt.render_frame(spp)
    px = pt.read_pixels()
    st = pt.stats()
    tri = pt.trace_primary()[0]
    hits = int((tri >= 0).sum())
    assert (px['n'] == spp).all() and np.isfinite(px['sum']).all() and (px['sum'] >= 0).all()
    assert st.samples == W * H * spp and st.raysPrimary == W * H * spp and st.raysOcclusion == 0 and st.raysIndirect == 0
    assert 0.9 < hits / (W * H) <= 1.0 and 0 < st.raysLight <= st.hitsPrimary and abs(st.hitsPrimary - hits * spp) < 0.01 * hits * spp
    band0, rows = 1000, 16
    o = make_oracle(flats[name], name, W, H, has_sky=False, seed=0x5EED)
    ref = oracle_py.new_pixels(rows * W)
    o.render_keyed(ref, band0 * W, spp, nthreads=8)
    got = px[band0 * W:(band0 + rows) * W]
    assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    assert ref['sum'].max() > 0
    del pt
    frame = np.zeros(W * H, P.PIXEL_DTYPE)
    for r in range(2):
        part = make_tracer(name, W, H, seed=0x5EED)
        part.render_bands(4, r, 2, spp)
        share = part.read_pixels()
        mine = P.band_rows(H, 4, r, 2)
        assert (share['n'].reshape(H, W)[mine] == spp).all() and share['n'].sum() == spp * len(mine) * W
        frame.reshape(H, W)[mine] = share.reshape(H, W)[mine]
        del part
    assert np.array_equal(frame['n'], px['n']) and np.array_equal(frame['sum'].view(np.uint32), px['sum'].view(np.uint32))


def test_two_handles_driven_from_two_host_threads(require_gpu):
    """The threading contract of the boundary: functions on one handle are not re-entrant, different handles may be
    driven from different threads (the reference's workers, main.d:207-215, each call computeOutputColor on their own
    slice).  Two handles render at once from two host threads, several calls each; each equals its solo render."""
    import threading
    specs = [('teapot', 640, 360, dict(has_sky=1, mode=0, seed=21)), ('chest1', 512, 288, dict(has_sky=1, mode=1, seed=22))]
    solo = []
    for name, W, H, cfg in specs:
        pt = make_tracer(name, W, H, **cfg)
        for _ in range(3):
            pt.render_frame(2)
        solo.append(pt.read_pixels()); del pt
    handles = [make_tracer(name, W, H, **cfg) for name, W, H, cfg in specs]
    errors = []
    start = threading.Barrier(2)

    def work(pt):
        try:
            start.wait()
            for _ in range(3):
                pt.render_frame(2)
            pt.synchronize()
        except Exception as e:                                        # reported below, in the test's thread
            errors.append(e)
    th = [threading.Thread(target=work, args=(pt,)) for pt in handles]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors
    for pt, ref in zip(handles, solo):
        got = pt.read_pixels()
        assert np.array_equal(got['n'], ref['n']) and np.array_equal(got['sum'].view(np.uint32), ref['sum'].view(np.uint32))


def test_headless_driver_image_and_project_file(require_gpu, flats):
    """The native caller (csrc/headless_main.cpp, the stand-in for main.d:142-345): loadScene, N passes of one sample
    per pixel with the reference's timing lines, tone map to a PPM, raw Pixel[] project file.  It ran as a child process
    at session start (conftest.pytest_sessionstart); its outputs equal the oracle's, byte for byte."""
    r = conftest.HEADLESS
    assert r is not None, 'the headless driver did not run (no /dev/kfd or pathtracer_amd/ptgpu_headless not built)'
    assert r['rc'] == 0, r['stdout']
    a = conftest.HEADLESS_ARGS
    W, H, passes = a['W'], a['H'], a['passes']
    out = r['stdout']
    assert 'Scene has 52 triangles' in out and 'Light 0 has 2 light triangles' in out                # scene.d:237, rendering.d:159
    assert all(('pass %d done in' % p) in out for p in range(1, passes + 1)) and 'Rendering took' in out   # main.d:279, 331
    o = make_oracle(flats[a['model']], a['model'], W, H, has_sky=False, seed=a['seed'])
    ref = oracle_py.new_pixels(W * H)
    o.render_keyed(ref, 0, passes, nthreads=8)
    proj = np.fromfile(r['project'], dtype=oracle_py.PIXEL_DTYPE)
    assert len(proj) == W * H and os.path.getsize(r['project']) == W * H * 28
    assert np.array_equal(proj['n'], ref['n']) and np.array_equal(proj['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    assert np.array_equal(proj['color'].view(np.uint32), ref['color'].view(np.uint32))
    raw = open(r['ppm'], 'rb').read()
    head = b'P6\n%d %d\n255\n' % (W, H)
    assert raw.startswith(head) and len(raw) == len(head) + W * H * 3
    rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(-1, 3)
    assert np.array_equal(rgb, oracle_py.tonemap(ref))
