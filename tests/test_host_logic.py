"""Host-side pieces of the product (no GPU): light table, camera state, flattened BVH, band partition."""
import numpy as np
import pytest

import oracle_py
import pathtracer_amd as P
from pathtracer_amd import multigpu
from conftest import CAMERAS, UP, make_oracle, model_path, soup

NAMES = sorted(CAMERAS)
COUNT_SHIFT, INDEX_MASK = 28, (1 << 28) - 1


def host_tracer(name=None, flat=None, **cfg):
    pt = P.PathTracer(64, 64, device=-1, **cfg)
    if name:
        pt.Scene(model_path(name))
    else:
        pt.upload_scene(flat['v'], flat['n'], flat['mat_idx'], flat['materials'])
    return pt


@pytest.mark.parametrize('name', NAMES)
def test_light_table_equals_oracle(name, flats):
    got = host_tracer(name).light_groups()
    ref = oracle_py.Oracle(flats[name], build_tree=False).light_groups()
    assert len(got) == len(ref)
    for (gr, ga), (rr, ra) in zip(got, ref):
        assert np.array_equal(gr.view(np.uint32), rr.view(np.uint32)) and ga == ra


def test_light_groups_one_per_emissive_material_in_triangle_order():
    """rendering.d:116-161: groups follow material order, members follow triangle order."""
    fl = soup(40, seed=1)
    fl['materials'] = np.array([[0, .7, .7, .7], [2.0, 1, 0, 0], [0, .1, .1, .1], [3.0, 0, 1, 0]], np.float32)
    fl['mat_idx'] = (np.arange(len(fl['v'])) % 4).astype(np.uint32)
    got = host_tracer(flat=fl).light_groups()
    ref = oracle_py.Oracle(fl, build_tree=False).light_groups()
    assert len(got) == 2
    for (gr, ga), (rr, ra) in zip(got, ref):
        assert np.array_equal(gr, rr) and ga == ra
        assert abs(gr[:, 10].sum() - 1) < 1e-5
    first_light_tri = fl['v'][1]
    assert np.array_equal(got[0][0][0, :3], first_light_tri[0])


@pytest.mark.parametrize('name', NAMES)
def test_camera_state_equals_oracle(name):
    for aspect in (1.0, 0.5625, 0.75):
        c = P.Camera(30.0, aspect).setTransform(CAMERAS[name][0], CAMERAS[name][1], UP).state
        r = oracle_py.camera_lookat(30.0, aspect, CAMERAS[name][0], CAMERAS[name][1], UP)
        assert np.array_equal(c.view(np.uint32), r.view(np.uint32))


def walk_bvh(nodes, order, root_ref):
    """Returns (leaf slots visited, max depth); asserts the structural invariants on the way."""
    seen = np.zeros(len(order), int)
    max_depth = 0
    stack = [(root_ref, None, 0)]
    while stack:
        ref, box, depth = stack.pop()
        max_depth = max(max_depth, depth)
        cnt, idx = ref >> COUNT_SHIFT, ref & INDEX_MASK
        if cnt:
            seen[idx:idx + cnt] += 1
            continue
        for k in (0, 1):
            n = nodes[2 * idx + k]
            cref = int(n[3:4].view(np.uint32)[0])
            cb = (n[0:3], n[4:7])
            if box is not None:                                  # child box inside parent box
                assert (cb[0] >= box[0] - 1e-6).all() and (cb[1] <= box[1] + 1e-6).all()
            stack.append((cref, cb, depth + 1))
    return seen, max_depth


def leaf_boxes(nodes, root_ref):
    out = []
    stack = [(root_ref, None)]
    while stack:
        ref, box = stack.pop()
        cnt, idx = ref >> COUNT_SHIFT, ref & INDEX_MASK
        if cnt:
            out.append((idx, cnt, box))
            continue
        for k in (0, 1):
            n = nodes[2 * idx + k]
            stack.append((int(n[3:4].view(np.uint32)[0]), (n[0:3], n[4:7])))
    return out


@pytest.mark.parametrize('name,leaf', [(n, l) for n in NAMES for l in (1, 4, 15)])
def test_bvh_structure(name, leaf, flats):
    pt = host_tracer(name, max_leaf_tris=leaf)
    nodes, order = pt.get_bvh()
    st = pt.stats()
    root_ref = int(nodes[0, 3:4].view(np.uint32)[0])
    v = flats[name]['v']
    degenerate = (v[:, 0] == v[:, 1]).all(1) | (v[:, 0] == v[:, 2]).all(1)
    n_leaf_tris = int((~degenerate).sum())
    order = order[:n_leaf_tris]
    seen, depth = walk_bvh(nodes, order, root_ref)
    assert (seen == 1).all()                                     # every leaf slot reachable exactly once
    assert sorted(order.tolist()) == np.flatnonzero(~degenerate).tolist()   # every non-degenerate triangle exactly once
    assert depth == st.bvhDepth and depth <= 60
    for idx, cnt, box in leaf_boxes(nodes, root_ref):
        assert 1 <= cnt <= leaf
        if box is not None:
            tv = v[order[idx:idx + cnt]].reshape(-1, 3)
            assert (tv >= box[0]).all() and (tv <= box[1]).all()  # padded box contains its triangles


@pytest.mark.parametrize('name', NAMES)
def test_bvh_traversal_finds_the_exhaustive_hit(name, flats):
    """The kernels' traversal order restated on the CPU, over the product's own flattened BVH, must accept the
    same triangle as the exhaustive search (closest hit) and agree on occlusion (any hit)."""
    W = H = 96
    pt = host_tracer(name)
    nodes, order = pt.get_bvh()
    root_ref = int(nodes[0, 3:4].view(np.uint32)[0])
    o = make_oracle(flats[name], name, W, H)
    rays = o.view_rays(0, W * H)
    n_nodes, n_tris, tri = o.bvh_count(nodes, root_ref, order, rays, any_hit=False)
    ref_tri = o.trace_primary(0, W * H, use_tree=False)[0]
    assert np.array_equal(tri, ref_tri)
    assert n_nodes >= 2 * W * H and n_tris < o.T * W * H / 4      # far fewer triangle tests than exhaustive
    rng = np.random.default_rng(7)
    lo, hi = flats[name]['v'].reshape(-1, 3).min(0), flats[name]['v'].reshape(-1, 3).max(0)
    org = rng.uniform(lo - 2, hi + 2, (4000, 3)); d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rr = np.concatenate([org, d], 1).astype(np.float32)
    _, _, tri_c = o.bvh_count(nodes, root_ref, order, rr, any_hit=False)
    assert np.array_equal(tri_c, o.trace_rays(rr, kind=0)[0])
    _, _, tri_a = o.bvh_count(nodes, root_ref, order, rr, any_hit=True)
    assert np.array_equal(tri_a >= 0, o.trace_rays(rr, kind=1)[0] >= 0)


def test_bvh_on_soup_and_degenerate_inputs():
    fl = soup(20000, seed=3)
    fl['v'][5] = fl['v'][5][[0, 0, 2]]                           # v0 == v1: can never be hit, left out of the tree
    fl['v'][6, 1, 0] = np.nan
    pt = host_tracer(flat=fl)
    nodes, order = pt.get_bvh()
    st = pt.stats()
    assert st.numTriangles == 20002
    order = order[:20000]
    assert 5 not in order and 6 not in order
    seen, depth = walk_bvh(nodes, order, int(nodes[0, 3:4].view(np.uint32)[0]))
    assert (seen == 1).all() and depth < 40


def test_empty_and_single_triangle_scenes():
    mats = np.array([[0, .7, .7, .7]], np.float32)
    pt = P.PathTracer(8, 8, device=-1)
    pt.upload_scene(np.zeros((0, 9), np.float32), np.zeros((0, 9), np.float32), np.zeros(0, np.uint32), mats)
    assert pt.stats().numTriangles == 0 and pt.stats().numLightGroups == 0
    one = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    pt.upload_scene(one, one, np.zeros(1, np.uint32), mats)
    nodes, order = pt.get_bvh()
    assert int(nodes[0, 3:4].view(np.uint32)[0]) >> COUNT_SHIFT == 1 and order[0] == 0


@pytest.mark.parametrize('height,band_rows,world', [(1080, 4, 8), (240, 4, 3), (37, 4, 2), (5, 4, 8), (2160, 16, 4)])
def test_band_partition_covers_every_row_once(height, band_rows, world):
    """main.d:184-192's bands dealt round-robin to ranks: a partition of the rows."""
    rows = [multigpu.rows_of_rank(height, band_rows, r, world) for r in range(world)]
    flat = sorted(x for r in rows for x in r)
    assert flat == list(range(height))
    sizes = [len(r) for r in rows]
    assert max(sizes) - min(sizes) <= band_rows
    with pytest.raises(ValueError):
        multigpu.band_share(world, world)


@pytest.mark.parametrize('name', NAMES)
def test_coverage_mask_is_conservative_and_tight(name, flats):
    """The per-pixel mask view rays are culled with (rendering.d:278-285 sample positions: the pixel's nominal
    position +- half a pixel): every sample position whose ray hits something -- front face by Scene.trace or any
    face by Scene.hitsNothing -- lies in a pixel whose bit is set, including the extreme jitter values; and the
    mask is not much larger than the set of pixels that can be hit."""
    W, H = 160, 96
    pt = P.PathTracer(W, H, device=-1)
    pt.Scene(model_path(name))
    cam = oracle_py.camera_lookat(30.0, H / W, CAMERAS[name][0], CAMERAS[name][1], UP)
    pt.set_camera(cam)
    mask = pt.coverage_mask()
    assert mask is not None and mask.shape == (W * H,)
    o = oracle_py.Oracle(flats[name], build_tree=False)
    pos, d, r, u = (cam[0:3].astype(np.float64), cam[3:6].astype(np.float64), cam[6:9].astype(np.float64), cam[9:12].astype(np.float64))
    tan_fov, aspect = float(cam[12]), float(cam[13])
    cols, rows = np.meshgrid(np.arange(W), np.arange(H))
    cols = cols.reshape(-1).astype(np.float64); rows = rows.reshape(-1).astype(np.float64)
    rng = np.random.default_rng(3)
    hit_any = np.zeros(W * H, bool)
    lim = 1.0 - 2.0 ** -24                       # uniform(-ps, ps) is [-ps, ps): largest value just below +ps
    jitters = [(-1.0, -1.0), (lim, lim), (-1.0, lim), (lim, -1.0), (0.0, 0.0)] + [tuple(rng.uniform(-1, lim, 2)) for _ in range(6)]
    for jx, jy in jitters:
        x = cols / W * 2 - 1 + jx / W
        y = rows / H * 2 - 1 + jy / H
        dirs = d[None, :] + r[None, :] * (x * tan_fov)[:, None] + u[None, :] * (y * tan_fov * aspect)[:, None]
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rays = np.concatenate([np.broadcast_to(pos, dirs.shape), dirs], axis=1).astype(np.float32)
        front = o.trace_rays(rays, kind=0)[0] >= 0
        anyf = o.trace_rays(rays, kind=1)[0] >= 0
        assert not (front & ~mask).any() and not (anyf & ~mask).any()
        hit_any |= front | anyf
    assert mask.sum() <= 1.6 * hit_any.sum() + 4 * (W + H)          # bounding rectangles of small triangles: close to the silhouette
    if name == 'teapot':
        assert mask.mean() < 0.5                                       # most of the frame is background


def test_coverage_mask_unusable_when_geometry_reaches_behind_the_camera():
    fl = soup(500, seed=2, extent=10.0)
    pt = P.PathTracer(64, 64, device=-1)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    pt.set_camera(oracle_py.camera_lookat(30.0, 1.0, (0.5, 0.5, 0.5), (5, 5, 5), UP))     # camera inside the soup
    assert pt.coverage_mask() is None
    pt.set_camera(oracle_py.camera_lookat(30.0, 1.0, (40, 30, 200), (0, 0, 0), UP))       # outside: everything in front
    m = pt.coverage_mask()
    assert m is not None and 0 < m.sum() < 64 * 64


def test_constant_sky_mask_never_hides_the_sun_disc_or_geometry():
    """constantSky pixels are skipped by raygen and filled with sampleSky's plain constant by accumulate
    (rendering.d:332-339): none of their sample positions may hit geometry or fall inside the sun disc
    dot(normalized(1,1,6), d) > 0.997."""
    W, H = 96, 64
    fl = soup(300, seed=4, extent=6.0)
    pt = P.PathTracer(W, H, device=-1)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    cam = oracle_py.camera_lookat(30.0, H / W, (-20, -20, -120), (0, 0, 0), UP)          # looks along (1, 1, 6): at the sun
    pt.set_camera(cam)
    covered, const_sky = pt.coverage_mask(constant_sky=True)
    assert covered is not None and not (covered & const_sky).any()
    pos, d, r, u = (cam[0:3].astype(np.float64), cam[3:6].astype(np.float64), cam[6:9].astype(np.float64), cam[9:12].astype(np.float64))
    tan_fov, aspect = float(cam[12]), float(cam[13])
    cols, rows = np.meshgrid(np.arange(W), np.arange(H))
    cols = cols.reshape(-1).astype(np.float64); rows = rows.reshape(-1).astype(np.float64)
    sun = np.array([1.0, 1.0, 6.0]) / np.sqrt(38.0)
    lim = 1.0 - 2.0 ** -24
    rng = np.random.default_rng(5)
    sunny = np.zeros(W * H, bool)
    for jx, jy in [(-1.0, -1.0), (lim, lim), (-1.0, lim), (lim, -1.0), (0.0, 0.0)] + [tuple(rng.uniform(-1, lim, 2)) for _ in range(8)]:
        x = cols / W * 2 - 1 + jx / W
        y = rows / H * 2 - 1 + jy / H
        dirs = d[None, :] + r[None, :] * (x * tan_fov)[:, None] + u[None, :] * (y * tan_fov * aspect)[:, None]
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        sunny |= dirs @ sun > 0.997 - 1e-6
    assert sunny.any() and not (sunny & const_sky).any()
    assert const_sky.sum() > 0.3 * W * H                                   # and it still removes most of the background
    # a camera that looks away from the sun: everything that is not covered is constant
    pt.set_camera(oracle_py.camera_lookat(30.0, H / W, (20, 20, 120), (0, 0, 0), UP))
    covered, const_sky = pt.coverage_mask(constant_sky=True)
    assert np.array_equal(const_sky, ~covered)


# ------------------------------------------------------------------ texcoord variant of Scene.trace (scene.d:745-796)
def interpolate_np(u, v, val0, val1, val2):
    """scene.d:68-105 in numpy float32, one operator per rounding."""
    f = np.float32
    u = np.maximum(f(u), f(0)); v = np.maximum(f(v), f(0))
    uv = f(u + v)
    if uv != 0:
        x = f(f(1) / uv); u1 = f(x * u); v1 = f(x * v)
    else:
        u1, v1 = u, v
    sqrt2 = f(1.414213562)
    d1 = f(np.sqrt(f(f(f(1) - u1) * f(f(1) - u1)) + f(v1 * v1)) / sqrt2)
    d2 = f(np.sqrt(f(f(u1 * u1) + f(f(f(1) - v1) * f(f(1) - v1)))) / sqrt2)
    ip = f(f(val1 * d1) + f(val2 * d2))
    if uv != 0:
        ln = f(np.sqrt(f(f(u1 * u1) + f(v1 * v1))))
        i1 = f(f(np.sqrt(f(f(u * u) + f(v * v)))) / ln)
    else:
        i1 = f(0)
    i2 = f(f(1) - i1)
    return f(f(val0 * i2) + f(ip * i1))


def one_triangle_scene():
    """One triangle in the z = 0 plane, counter-clockwise seen from +z, with distinct texcoords per vertex."""
    v = np.array([[[0, 0, 0], [4, 0, 0], [0, 4, 0]]], np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (1, 3, 1))
    fl = {'v': v, 'n': n, 'mat_idx': np.zeros(1, np.uint32), 'materials': np.array([[0, .5, .5, .5]], np.float32),
          'tex': np.array([[[0.25, 0.5], [1.0, 0.0], [0.0, 2.0]]], np.float32)}
    return fl


def test_oracle_texcoord_trace_known_answers():
    """Corners return the vertex values (u weights vertex 2, v vertex 1), back faces are reported or ignored,
    and interior points follow the numpy restatement of interpolate()."""
    fl = one_triangle_scene()
    o = oracle_py.Oracle(fl, build_tree=False)
    down = lambda x, y: [x, y, 5, 0, 0, -1]
    up = lambda x, y: [x, y, -5, 0, 0, 1]
    rays = np.array([down(0, 0), down(4, 0), down(0, 4), down(1, 1), down(0.5, 2.5), down(9, 9), up(1, 1)], np.float32)
    res, tri, t, tc = o.trace_tex(rays, ignore_backfaces=False)
    sides = res[[0, 3, 6]]
    assert set(sides[:2]) == {sides[0]} and sides[2] == 1 - sides[0]      # one side front, the other back
    assert res[5] == 2 and tri[5] == -1 and t[5] > 3e38 and np.array_equal(tc[5], [0, 0])
    assert np.array_equal(tri[[0, 1, 2, 3, 4, 6]], np.zeros(6, np.int32)) and np.allclose(t[[0, 1, 2, 3, 4, 6]], 5)
    tex = fl['tex'][0]
    assert np.allclose(tc[0], tex[0], atol=1e-6)
    # which corner u and v weight is fixed by the intersection routine; both orders of (4,0) and (0,4) are vertex values
    assert {tuple(np.round(tc[1], 5)), tuple(np.round(tc[2], 5))} == {tuple(tex[1]), tuple(tex[2])}
    res_f, tri_f, t_f, tc_f = o.trace_tex(rays, ignore_backfaces=True)
    back = res == 1
    assert (res_f[back] == 2).all() and (tri_f[back] == -1).all() and np.array_equal(res_f[~back], res[~back])
    # interior: recompute from the oracle's own (u, v) of the plain trace
    front_rays = rays[res == 0]
    otri, ot, ou, ov = o.trace_rays(front_rays, kind=0, use_tree=False)
    for i in range(len(front_rays)):
        for c in range(2):
            e = interpolate_np(ou[i], ov[i], tex[0][c], tex[1][c], tex[2][c])
            assert tc[res == 0][i][c] == e


def test_texcoord_trace_agrees_with_plain_trace_on_front_faces(flats):
    """With back faces ignored the variant finds exactly the hits of Scene.trace (scene.d:681-743)."""
    from test_gpu_parity import random_rays
    fl = flats['teapot']
    o = oracle_py.Oracle(fl, build_tree=False)
    rays = random_rays(fl, 3000, 9)
    res, tri, t, tc = o.trace_tex(rays, ignore_backfaces=True)
    otri, ot, ou, ov = o.trace_rays(rays, kind=0, use_tree=False)
    assert np.array_equal(tri, otri) and np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(res == 0, otri >= 0) and not (res == 1).any()
    res2, tri2, t2, _ = o.trace_tex(rays, ignore_backfaces=False)
    assert (res2 == 1).sum() > 100 and (t2 <= t).all()


@pytest.mark.parametrize('name', NAMES)
def test_flattened_texcoords_equal_reader_restatement(name, flats):
    """TriangleData.tex (scene.d:305-307) as the product's reader flattens it vs the Python restatement of the reader."""
    got = host_tracer(name).get_texcoords()
    assert np.array_equal(got.view(np.uint32), flats[name]['tex'].astype(np.float32).view(np.uint32))
    assert np.abs(got).max() > 0


def test_texcoords_of_uploaded_scenes_default_to_zero_and_can_be_set():
    fl = soup(50, seed=3)
    pt = host_tracer(flat=fl)
    assert not pt.get_texcoords().any()
    tex = np.random.default_rng(0).random((len(fl['v']), 3, 2)).astype(np.float32)
    pt.set_texcoords(tex)
    assert np.array_equal(pt.get_texcoords(), tex)
    with pytest.raises(ValueError):
        pt.set_texcoords(tex[:10])
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])      # a new scene drops them
    assert not pt.get_texcoords().any()


@pytest.mark.parametrize('name', NAMES + ['soup'])
def test_quantised_nodes_contain_their_boxes(name, flats):
    """Scenes traversed from global memory get 16-byte nodes on a 16-bit grid over the root box: every box returned for
    such a scene contains the float box of the same node, by no more than a few grid cells, and the tree is the same."""
    if name == 'soup':
        fl = soup(3000, seed=11); a = host_tracer(flat=fl, flags=P.PT_FLAG_NO_TREELETS); b = host_tracer(flat=fl, flags=P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_NO_TREELETS)
    else:
        a = host_tracer(name); b = host_tracer(name, flags=P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_NO_TREELETS)
    assert b.stats().sceneInLds == 0
    (na, oa), (nb, ob) = a.get_bvh(), b.get_bvh()
    assert np.array_equal(oa, ob) and np.array_equal(na[:, 3].view(np.uint32), nb[:, 3].view(np.uint32))
    real = na[:, 4:7] >= na[:, 0:3]                      # padding nodes are empty boxes
    real = real.all(1)
    real[1] = False                                       # node 1 only keeps sibling pairs aligned
    mn_a, mx_a, mn_b, mx_b = na[real, 0:3], na[real, 4:7], nb[real, 0:3], nb[real, 4:7]
    assert (mn_b <= mn_a).all() and (mx_b >= mx_a).all()
    cell = (na[0, 4:7] - na[0, 0:3]) / 65529.0
    assert ((mn_a - mn_b) <= 3.01 * cell + 1e-30).all() and ((mx_b - mx_a) <= 3.01 * cell + 1e-30).all()


@pytest.mark.parametrize('name', ['teapot', 'soup'])
def test_cache_line_layout_of_the_node_pairs_keeps_the_tree(name, flats):
    """Scenes traversed from global memory: sibling pairs are re-ordered so that a 128-byte line (four pairs of 16-byte
    nodes) holds a pair together with descendants.  The tree is the same -- walked from the root in step with the
    builder's order, every box and every leaf agrees -- most steps down land in the line of the pair they come from,
    and few slots are padding."""
    if name == 'soup':
        fl = soup(20000, seed=5)
        a = host_tracer(flat=fl, flags=P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_NO_TREELETS); b = host_tracer(flat=fl, flags=P.PT_FLAG_NO_LDS_SCENE)
    else:
        a = host_tracer(name, flags=P.PT_FLAG_NO_LDS_SCENE | P.PT_FLAG_NO_TREELETS); b = host_tracer(name, flags=P.PT_FLAG_NO_LDS_SCENE)
    (na, oa), (nb, ob) = a.get_bvh(), b.get_bvh()
    assert np.array_equal(oa, ob)
    ra, rb = na[:, 3].view(np.uint32), nb[:, 3].view(np.uint32)
    assert len(nb) % 8 == 0 and len(nb) >= len(na)
    stack = [(int(ra[0]), int(rb[0]), 0)]
    pairs = same_line = 0
    seen = set()
    while stack:
        x, y, parent = stack.pop()
        assert (x >> 28) == (y >> 28)
        if x >> 28:
            assert x == y                                            # same leaf: same triangles
            continue
        assert y not in seen; seen.add(y)
        pairs += 1; same_line += (y // 4 == parent // 4)
        for k in (0, 1):
            assert np.array_equal(na[2 * x + k, [0, 1, 2, 4, 5, 6]], nb[2 * y + k, [0, 1, 2, 4, 5, 6]])
            stack.append((int(ra[2 * x + k]), int(rb[2 * y + k]), y))
    assert pairs == len(na) // 2 - 1                                 # every pair of the original, once
    unused = rb.reshape(-1, 2)[:, 0] == 0xFFFFFFFF
    assert int((~unused).sum()) == pairs + 1 and unused.mean() < 0.2
    assert same_line / pairs > 0.5                                   # at most 3 of 4: the first pair of a line comes from another
