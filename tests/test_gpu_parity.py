"""GPU parity of the intersection path, through the C ABI, against the CPU oracle and the golden vectors.

Bar: triangle ids bit-exact; t, u, v within 1e-5 (they are in fact bit-equal: both sides evaluate the same
fixed sequence of fp32 operations); Scene.hitsNothing equal as booleans."""
import os

import numpy as np
import pytest

import oracle_py
import pathtracer_amd as P
import conftest
from conftest import CAMERAS, GOLDEN, make_oracle, make_tracer, model_path, soup

pytestmark = pytest.mark.gpu
NAMES = sorted(CAMERAS)
TOL = 1e-5


def assert_hits_equal(got, ref):
    tri, t, u, v = got; otri, ot, ou, ov = ref
    assert np.array_equal(tri, otri)                                  # integer ids: bit-exact
    hit = otri >= 0
    assert np.all(np.abs(t[hit] - ot[hit]) <= TOL * np.abs(ot[hit]))  # relative for t
    assert np.all(np.abs(u[hit] - ou[hit]) <= TOL) and np.all(np.abs(v[hit] - ov[hit]) <= TOL)
    assert np.array_equal(t[hit].view(np.uint32), ot[hit].view(np.uint32))   # stronger: same bits
    assert np.array_equal(u[hit].view(np.uint32), ou[hit].view(np.uint32))
    assert np.array_equal(v[hit].view(np.uint32), ov[hit].view(np.uint32))
    assert (t[~hit] > 3e38).all()


def test_rng_words_equal_oracle(require_gpu):
    pt = P.PathTracer(8, 8, seed=2024)
    g = np.load(os.path.join(GOLDEN, 'oracle_teapot_64x64.npz'))
    assert np.array_equal(pt.rng_stream(17, 3, 1, 16), g['rng_words'])
    for pixel, sample, stream in ((0, 0, 0), (123456, 77, 5), (2 ** 31 - 1, 2 ** 20, 1)):
        assert np.array_equal(pt.rng_stream(pixel, sample, stream, 256), oracle_py.keyed_stream(2024, pixel, sample, stream, 256))


@pytest.mark.parametrize('name,W,H', [('cornell-box', 512, 512), ('chest1', 640, 360), ('teapot', 1920, 1080)])
def test_primary_hits_equal_exhaustive_oracle(require_gpu, name, W, H, flats):
    """Full-frame probe without jitter (teapot at the 1920x1080 of the benchmark configuration)."""
    pt = make_tracer(name, W, H)
    o = make_oracle(flats[name], name, W, H)
    assert np.array_equal(pt.view_rays().view(np.uint32), o.view_rays(0, W * H).view(np.uint32))   # rays bit-exact
    got = pt.trace_primary()
    ref = o.trace_primary(0, W * H, use_tree=False)
    assert_hits_equal(got, ref)
    assert (ref[0] >= 0).sum() > 0.15 * W * H


@pytest.mark.parametrize('name', NAMES)
def test_primary_hits_equal_golden(require_gpu, name):
    g = np.load(os.path.join(GOLDEN, 'oracle_%s_64x64.npz' % name))
    pt = make_tracer(name, 64, 64)
    assert_hits_equal(pt.trace_primary(), (g['tri'], g['t'], g['u'], g['v']))


def test_primary_window_and_offsets(require_gpu, flats):
    W, H = 200, 120
    pt = make_tracer('teapot', W, H); o = make_oracle(flats['teapot'], 'teapot', W, H)
    for off, cnt in ((0, 1), (W * H - 1, 1), (777, 5003), (0, 0)):
        got = pt.trace_primary(off, cnt)
        if cnt:
            assert_hits_equal(got, o.trace_primary(off, cnt, use_tree=False))
        else:
            assert len(got[0]) == 0
    with pytest.raises(P.PtError):
        pt.trace_primary(W * H - 2, 3)


def random_rays(flat, n, seed):
    rng = np.random.default_rng(seed)
    pts = flat['v'].reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    org = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], 1).astype(np.float32)
    # a share of rays that start ON geometry (shadow / bounce rays do): origin = point on a triangle + tiny offset
    k = n // 3
    tri = rng.integers(0, len(flat['v']), k)
    b = rng.dirichlet([1, 1, 1], k).astype(np.float32)
    p = (flat['v'][tri] * b[:, :, None]).sum(1)
    rays[:k, :3] = p + rays[:k, 3:] * np.float32(1e-5)
    return rays


@pytest.mark.parametrize('name', NAMES)
def test_scene_trace_and_hits_nothing_on_random_rays(require_gpu, name, flats):
    """Scene.trace (scene.d:681-743) and Scene.hitsNothing (scene.d:801-838) for arbitrary rays, inside and outside."""
    pt = make_tracer(name, 16, 16)
    o = make_oracle(flats[name], name, 16, 16)
    rays = random_rays(flats[name], 60000, 42)
    tri, t, u, v, nrm = pt.trace(rays)
    ref = o.trace_rays(rays, kind=0, use_tree=False)
    assert_hits_equal((tri, t, u, v), ref)
    hit = np.flatnonzero(ref[0] >= 0)
    assert len(hit) > 1000
    for i in hit[:: max(1, len(hit) // 300)]:                          # interpolated shading normal, scene.d:711-723
        en = o.interp_normal(ref[0][i], ref[2][i], ref[3][i])
        assert np.allclose(nrm[i], en, atol=2e-6, equal_nan=True)
    free = pt.hitsNothing(rays)
    ofree = o.trace_rays(rays, kind=1, use_tree=False)[0] < 0
    assert np.array_equal(free, ofree)
    assert 0 < free.sum() < len(rays)


@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('lds', [True, False])
def test_texcoord_trace_equals_oracle(require_gpu, name, lds, flats):
    """Scene.trace's texcoord variant (scene.d:745-796) for both IgnoreBackfaces values: TraceResult and triangle
    ids bit-exact, rayPos and the interpolated texcoords bit-equal (same fixed fp32 sequence on both sides)."""
    pt = make_tracer(name, 16, 16, flags=0 if lds else P.PT_FLAG_NO_LDS_SCENE)
    assert pt.stats().sceneInLds == (1 if lds else 0)
    o = make_oracle(flats[name], name, 16, 16)
    rays = random_rays(flats[name], 30000, 7)
    for ignore in (True, False):
        res, tri, t, tc = pt.trace_tex(rays, ignore_backfaces=ignore)
        ores, otri, ot, otc = o.trace_tex(rays, ignore_backfaces=ignore)
        assert np.array_equal(res, ores) and np.array_equal(tri, otri)
        hit = ores != 2
        assert np.all(np.abs(t[hit] - ot[hit]) <= TOL * np.abs(ot[hit])) and np.all(np.abs(tc - otc) <= TOL)
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
        assert np.array_equal(tc.view(np.uint32), otc.view(np.uint32))
        assert (res == 0).sum() > 500 and ((res == 1).sum() > 500) == (not ignore)
        assert np.abs(tc[hit]).max() > 0
    # caller-supplied texcoords replace the file's
    tex = np.random.default_rng(1).random(flats[name]['tex'].shape).astype(np.float32)
    pt.set_texcoords(tex)
    fl2 = dict(flats[name]); fl2['tex'] = tex
    o2 = oracle_py.Oracle(fl2, build_tree=False)
    res, tri, t, tc = pt.trace_tex(rays[:5000], ignore_backfaces=False)
    ores, otri, ot, otc = o2.trace_tex(rays[:5000], ignore_backfaces=False)
    assert np.array_equal(res, ores) and np.array_equal(tc.view(np.uint32), otc.view(np.uint32))
    assert len(pt.trace_tex(np.zeros((0, 6), np.float32))[0]) == 0


def test_scene_traversed_from_global_memory_gives_the_same_hits(require_gpu, flats):
    """PT_FLAG_NO_LDS_SCENE: the HBM/L2 variant of every traversal kernel against the LDS-staged one."""
    name, W, H = 'teapot', 320, 180
    a = make_tracer(name, W, H)
    b = make_tracer(name, W, H, flags=P.PT_FLAG_NO_LDS_SCENE)
    assert a.stats().sceneInLds == 1 and b.stats().sceneInLds == 0
    for x, y in zip(a.trace_primary(), b.trace_primary()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    rays = random_rays(flats[name], 20000, 1)
    assert np.array_equal(a.hitsNothing(rays), b.hitsNothing(rays))


@pytest.mark.parametrize('leaf', [1, 2, 8, 15])
def test_hits_do_not_depend_on_the_tree(require_gpu, leaf, flats):
    name, W, H = 'teapot', 256, 144
    ref = make_oracle(flats[name], name, W, H).trace_primary(0, W * H, use_tree=False)
    pt = make_tracer(name, W, H, max_leaf_tris=leaf)
    assert_hits_equal(pt.trace_primary(), ref)


def test_soup_scene_from_global_memory(require_gpu):
    """200k-triangle soup (too large for LDS): closest hits and occlusion against the exhaustive oracle on a sample
    of camera rays and random interior rays."""
    fl = soup(200000, seed=0x5EED)
    W, H = 96, 54
    pt = P.PathTracer(W, H)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
    pt.set_camera(cam)
    assert pt.stats().sceneInLds == 0
    o = oracle_py.Oracle(fl, build_tree=False); o.configure(width=W, height=H); o.set_camera(cam.state)
    assert_hits_equal(pt.trace_primary(), o.trace_primary(0, W * H, use_tree=False))
    rays = random_rays(fl, 3000, 5)
    tri, t, u, v, _ = pt.trace(rays)
    assert_hits_equal((tri, t, u, v), o.trace_rays(rays, kind=0))
    assert np.array_equal(pt.hitsNothing(rays), o.trace_rays(rays, kind=1)[0] < 0)


def check_tree(nodes, order, num_tris, v, max_leaf):
    """Structure of a flattened BVH (layout of bvh.h): every hittable triangle in exactly one leaf, every box
    contains what hangs below it, leaves within the size bound.  Returns (depth, number of leaves)."""
    refs = nodes[:, 3].view(np.uint32)
    seen = np.zeros(len(order), np.int32)
    vv = v.reshape(-1, 3, 3)
    stack = [(0, 1)]
    depth = leaves = 0
    while stack:
        node, d = stack.pop()
        depth = max(depth, d)
        ref = int(refs[node]); cnt, idx = ref >> 28, ref & ((1 << 28) - 1)
        mn, mx = nodes[node, 0:3], nodes[node, 4:7]
        if cnt == 0:
            for c in (2 * idx, 2 * idx + 1):
                assert (nodes[c, 0:3] >= mn - 1e-6).all() and (nodes[c, 4:7] <= mx + 1e-6).all()
                stack.append((c, d + 1))
        else:
            assert cnt <= max_leaf
            leaves += 1
            seen[idx:idx + cnt] += 1
            tri = vv[order[idx:idx + cnt]]
            assert (tri.min((0, 1)) >= mn).all() and (tri.max((0, 1)) <= mx).all()
    assert (seen == 1).all() and len(order) == num_tris
    return depth, leaves


def test_bvh_built_on_the_gpu(require_gpu, flats):
    """PT_FLAG_GPU_BVH (the default from 1 M triangles up): Morton-order LBVH built by HIP kernels.  The tree is a
    valid BVH in the layout the traversal kernels read, and -- the closest hit being independent of the tree -- view
    rays, random rays and a rendered image are the same bits as with the SAH tree built on the host."""
    fl = soup(200000, seed=0x5EED)
    W, H = 96, 54
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
    tr = []
    for flags in (P.PT_FLAG_GPU_BVH, P.PT_FLAG_HOST_BVH):
        pt = P.PathTracer(W, H, has_sky=1, seed=9, flags=flags)
        pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
        pt.set_camera(cam)
        tr.append(pt)
    g, h = tr
    assert g.stats().bvhOnGpu == 1 and h.stats().bvhOnGpu == 0
    nodes, order = g.get_bvh()
    depth, leaves = check_tree(nodes.reshape(-1, 8), order, len(fl['v']), fl['v'], 4)
    assert depth - 1 == g.stats().bvhDepth <= 60 and leaves > len(order) / 4          # bvhDepth counts edges, the root is 0
    for x, y in zip(g.trace_primary(), h.trace_primary()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    rays = random_rays(fl, 20000, 7)
    for x, y in zip(g.trace(rays)[:4], h.trace(rays)[:4]):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.array_equal(g.hitsNothing(rays), h.hitsNothing(rays))
    g.render_frame(4); h.render_frame(4)
    assert np.array_equal(g.read_pixels()['sum'].view(np.uint32), h.read_pixels()['sum'].view(np.uint32))
    # a scene with duplicate centroids and degenerate / NaN triangles among the hittable ones
    v = np.concatenate([fl['v'][:3000], fl['v'][:3000], np.zeros((50, 3, 3), np.float32),
                        np.full((5, 3, 3), np.nan, np.float32)]).astype(np.float32)
    t2 = P.PathTracer(W, H, flags=P.PT_FLAG_GPU_BVH)
    t2.upload_scene(v, np.zeros_like(v), np.zeros(len(v), np.uint32), fl['materials'])
    t2.set_camera(cam)
    assert t2.stats().bvhOnGpu == 1
    nodes, order = t2.get_bvh()
    check_tree(nodes.reshape(-1, 8), order[:6000], 6000, v, 4)          # the 55 triangles that cannot be hit are left out
    o = oracle_py.Oracle({'v': v, 'n': np.zeros_like(v), 'mat_idx': np.zeros(len(v), np.uint32), 'materials': fl['materials']},
                         build_tree=False)
    r2 = random_rays({'v': v[:6000]}, 4000, 3)
    tri, t, u, vv, _ = t2.trace(r2)
    assert_hits_equal((tri, t, u, vv), o.trace_rays(r2, kind=0))
    # the shipped teapot through the GPU builder
    a = make_tracer('teapot', 256, 144, flags=P.PT_FLAG_GPU_BVH)
    assert a.stats().bvhOnGpu == 1
    assert_hits_equal(a.trace_primary(), make_oracle(flats['teapot'], 'teapot', 256, 144).trace_primary(0, 256 * 144, use_tree=False))


def test_degenerate_inputs(require_gpu):
    mats = np.array([[0, .7, .7, .7]], np.float32)
    pt = P.PathTracer(32, 32)
    pt.set_camera(P.Camera(30.0, 1.0).setTransform((0, -5, 0), (0, 0, 0), (0, 0, 1)))
    pt.upload_scene(np.zeros((0, 9), np.float32), np.zeros((0, 9), np.float32), np.zeros(0, np.uint32), mats)   # empty scene
    tri, t, _, _ = pt.trace_primary()
    assert (tri == -1).all() and (t > 3e38).all()
    assert pt.hitsNothing(np.array([[0, -5, 0, 0, 1, 0]], np.float32)).all()
    # one front-facing triangle, one zero-area triangle, one triangle with a NaN vertex, one back-facing triangle
    v = np.array([[-1, 0, -1, 0, 0, 1, 1, 0, -1],            # normal +y?  (v1-v0)x(v2-v0) = (1,0,2)x(2,0,0) = (0,4,0)... faces +y
                  [0, 1, 0, 0, 1, 0, 0, 1, 0],
                  [np.nan, 2, 0, 1, 2, 0, 0, 2, 1],
                  [-1, 3, -1, 1, 3, -1, 0, 3, 1]], np.float32)
    pt.upload_scene(v, v, np.zeros(4, np.uint32), mats)
    o = oracle_py.Oracle({'v': v, 'n': v, 'mat_idx': np.zeros(4, np.uint32), 'materials': mats}, build_tree=False)
    rays = np.array([[0, -5, 0, 0, 1, 0], [0, 10, 0, 0, -1, 0], [0, -5, 0, 0, -1, 0], [0, -5, 0, np.nan, 1, 0],
                     [0, 0, 0, 0, 1, 0], [0.2, -5, 0.1, 0, 1, 0]], np.float32)
    tri, t, u, v2, _ = pt.trace(rays)
    assert_hits_equal((tri, t, u, v2), o.trace_rays(rays, kind=0))
    assert np.array_equal(pt.hitsNothing(rays), o.trace_rays(rays, kind=1)[0] < 0)
    assert set(tri.tolist()) <= {-1, 0, 3}


def test_visit_counters_match_cpu_restatement_of_the_traversal(require_gpu, flats):
    """PT_FLAG_COUNT_VISITS (the per-ray node / triangle figures of the roofline) against the oracle walking the
    same flattened BVH in the same order.  The kernels use the hardware reciprocal in the slab test, the oracle
    1/x, so a box can flip at the margin: agreement to 0.1 %."""
    name, W, H = 'teapot', 320, 180
    pt = make_tracer(name, W, H, flags=P.PT_FLAG_COUNT_VISITS)
    o = make_oracle(flats[name], name, W, H)
    nodes, order = pt.get_bvh()
    root = int(nodes[0, 3:4].view(np.uint32)[0])
    pt.reset_stats(); pt.trace_primary()
    st = pt.stats()
    n_nodes, n_tris, _ = o.bvh_count(nodes, root, order, o.view_rays(0, W * H))
    assert abs(st.nodesClosest - n_nodes) <= 1e-3 * n_nodes
    assert abs(st.trisClosest - n_tris) <= 1e-3 * n_tris
    # any-hit queries: children in storage order when the scene is in LDS, near child first when it is not
    rays = random_rays(flats[name], 40000, 3)
    for flags, order_rule in ((P.PT_FLAG_COUNT_VISITS, 2), (P.PT_FLAG_COUNT_VISITS | P.PT_FLAG_NO_LDS_SCENE, 1)):
        pa = make_tracer(name, W, H, flags=flags)
        nodes, order = pa.get_bvh()
        root = int(nodes[0, 3:4].view(np.uint32)[0])             # (the node pairs of a scene in global memory are laid out anew)
        pa.reset_stats(); free = pa.hitsNothing(rays)
        st = pa.stats()
        n_nodes, n_tris, tri = o.bvh_count(nodes, root, order, rays, any_hit=order_rule)
        assert np.array_equal(free, tri < 0)
        assert abs(st.nodesAny - n_nodes) <= 1e-3 * n_nodes and abs(st.trisAny - n_tris) <= 1e-3 * n_tris


@pytest.mark.parametrize('extent,size,offset', [(1e4, 1e-2, 0.0), (5.0, 3.0, 1e5), (1e-3, 4e-4, 0.0), (100.0, 40.0, -3e3)])
def test_quantised_nodes_on_extreme_scales(require_gpu, extent, size, offset):
    """Scenes traversed from global memory (16-bit box grid over the root box) whose scale defeats naive rounding:
    triangles far smaller than a grid cell, coordinates far from the origin (the grid-to-world map is evaluated in
    fp32), a scene of millimetre extent, triangles as large as the scene.  Hits must equal the exhaustive oracle's."""
    fl = soup(2500, seed=3, extent=extent, size=size)
    fl['v'] = (fl['v'] + np.float32(offset)).astype(np.float32)
    pt = P.PathTracer(16, 16, flags=P.PT_FLAG_NO_LDS_SCENE, eps=1e-5 if extent > 1 else 1e-12)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    assert pt.stats().sceneInLds == 0
    o = oracle_py.Oracle(fl, build_tree=False); o.configure(width=16, height=16, eps=1e-5 if extent > 1 else 1e-12)
    rays = random_rays(fl, 12000, 17)
    tri, t, u, v, _ = pt.trace(rays)
    ref = o.trace_rays(rays, kind=0, use_tree=False)
    assert_hits_equal((tri, t, u, v), ref)
    assert np.array_equal(pt.hitsNothing(rays), o.trace_rays(rays, kind=1, use_tree=False)[0] < 0)
    assert (ref[0] >= 0).sum() > 50


def test_direction_helpers_equal_oracle_bit_for_bit(require_gpu):
    """angleToDirection / angleToLocalDirection (rendering.d:309-330) on the device against the oracle for the angle
    ranges the integrator draws (psi in [0, 2 pi), phi in [0, pi/2] and the sun's [0.99 pi/2, pi/2]) and normals
    on both sides of the |n.x| > 0.9 switch of the helper frame: the same bits, sin and cos included."""
    rng = np.random.default_rng(11)
    n = 200000
    psi = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    phi = np.where(rng.random(n) < 0.5, rng.uniform(0, np.pi / 2, n), rng.uniform(0.99 * np.pi / 2, np.pi / 2, n)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm[: n // 4, 0] *= 8                                        # many normals with |n.x| > 0.9 after normalisation
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    nrm[::5] = 0                                                 # rows for angleToLocalDirection
    edge = np.array([[0, 0], [np.pi / 2, 0], [0, np.pi], [np.pi / 2, 2 * np.pi], [np.pi / 4, np.pi / 2]], np.float32)
    phi[:5], psi[:5] = edge[:, 0], edge[:, 1]
    pt = P.PathTracer(8, 8, device=0)
    got = pt.angle_directions(phi, psi, nrm)
    ref = oracle_py.angle_directions(phi, psi, nrm)
    assert (np.abs(nrm[:, 0]) > 0.9).sum() > 1000
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    unit = np.linalg.norm(got[np.abs(nrm).sum(1) > 0].astype(np.float64), axis=1)
    assert np.abs(unit - 1).max() < 1e-5


@pytest.mark.parametrize('case', ['cornell-box', 'teapot', 'chest1', 'soup', 'soup-behind'])
def test_view_ray_masks_built_on_the_gpu_equal_the_host_builder(require_gpu, case):
    """The masks rendering uses are rasterised on the GPU (mask_gpu.hip); the host builder (scene_host.cpp, checked
    against the oracle for conservativeness in test_host_logic.py) computes the same arithmetic: same bits, for frames
    whose width is and is not a multiple of 32, big and small triangles, and a camera with geometry behind it."""
    from conftest import UP
    for W, H in ((160, 90), (333, 217)):
        dev = P.PathTracer(W, H, device=0, has_sky=1)
        host = P.PathTracer(W, H, device=-1, has_sky=1)
        if case.startswith('soup'):
            fl = soup(20000, seed=9, extent=30.0, size=1.5)
            for t in (dev, host):
                t.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
            frm = (90, 70, 60) if case == 'soup' else (10, 5, 0)         # the second camera sits inside the soup
            cam = P.Camera(30.0, H / W).setTransform(frm, (0, 0, 0), UP)
        else:
            for t in (dev, host):
                t.Scene(conftest.model_path(case))
            cam = P.Camera(30.0, H / W).setTransform(CAMERAS[case][0], CAMERAS[case][1], UP)
        dev.set_camera(cam); host.set_camera(cam)
        a, b = dev.coverage_mask(constant_sky=True), host.coverage_mask(constant_sky=True)
        if case == 'soup-behind':
            assert a == (None, None) and b == (None, None)
            dev.render_frame(1); assert (dev.read_pixels()['n'] == 1).all()           # renders with the box cull instead
            continue
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[0].any() and (a[1].any() or case == "cornell-box")         # the cornell room fills the frame
        dev.render_frame(1)
        assert dev.stats().maskBuildMs >= 0.0


@pytest.mark.parametrize('lds', [True, False])
def test_rays_that_start_far_from_the_scene(require_gpu, lds, flats):
    """Origins 10^2 and 10^3 scene extents away: the triangle test subtracts the origin in fp32, so it answers for a ray
    whose origin is off by an ulp of those large coordinates; the box test widens by that much (RayInv in kernels.hip)
    instead of relying on the scene-relative padding of the boxes, so the traversal still finds exactly what the
    exhaustive oracle accepts.  At 10^4 extents the triangle test itself (scene.d:60-105 restated) accepts triangles
    that the exact ray misses by a tenth of the scene -- cancellation in its dot products -- so a traversal that visits
    only the boxes near the ray cannot agree on every ray; there the agreement is counted, not required to be total."""
    fl = flats['teapot']
    pt = P.PathTracer(64, 64, device=0, flags=0 if lds else P.PT_FLAG_NO_LDS_SCENE)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    assert pt.stats().sceneInLds == (1 if lds else 0)
    o = oracle_py.Oracle(fl, build_tree=False)
    rng = np.random.default_rng(8)
    n = 20000
    centre = np.array([1.0, 0.0, 7.0]); extent = 17.0
    for far in (1e2, 1e3, 1e4):
        dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        org = centre - dirs * extent * far + rng.normal(size=(n, 3)) * extent * 0.3
        target = centre + rng.uniform(-0.5, 0.5, size=(n, 3)) * extent
        d = target - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([org, d], axis=1).astype(np.float32)
        tri, t, u, v, _ = pt.trace(rays)
        otri, ot, ou, ov = o.trace_rays(rays, kind=0)
        free = pt.hitsNothing(rays); ofree = o.trace_rays(rays, kind=1)[0] < 0
        assert (otri >= 0).mean() > 0.5
        if far <= 1e3:
            assert np.array_equal(tri, otri)
            hit = otri >= 0
            assert np.array_equal(t[hit].view(np.uint32), ot[hit].view(np.uint32))
            assert np.array_equal(free, ofree)
        else:
            assert (tri != otri).mean() < 1e-3 and (free != ofree).mean() < 1e-3
