"""The CPU oracle against what can be known about the reference without running it: the asserts its source
holds (scene.d:347, 383-384, 498; rendering.d:139, 421-423), known answers derived from its data files, its own
internal consistency (sphere-tree traversal vs exhaustive search) and the committed golden vectors."""
import os

import numpy as np
import pytest

import oracle_py
import thmodel_ref
from conftest import CAMERAS, GOLDEN, UP, make_oracle, model_path

NAMES = sorted(CAMERAS)


@pytest.mark.parametrize('name', NAMES)
def test_sphere_tree_invariants(name, flats):
    o = oracle_py.Oracle(flats[name], build_tree=True)
    assert o.check_tree() == 0                                   # r^2 > 0, one rooted binary tree, children inside at creation
    assert oracle_py.lib().orc_num_nodes(o.h) == 2 * o.T - 1     # scene.d:329: 2T allocated, 2T-1 used


@pytest.mark.parametrize('name', NAMES)
def test_tree_traversal_equals_exhaustive_search(name, flats):
    W = H = 96
    o = make_oracle(flats[name], name, W, H, build_tree=True)
    tri_b, t_b, u_b, v_b = o.trace_primary(0, W * H, use_tree=False)
    tri_t, t_t, u_t, v_t = o.trace_primary(0, W * H, use_tree=True)
    assert np.array_equal(t_b, t_t)                              # same closest distance everywhere
    diff = tri_b != tri_t                                        # ids may differ only where two triangles tie exactly
    if diff.any():
        rays = o.view_rays(0, W * H)[diff]
        for r, a, b in zip(rays, tri_b[diff], tri_t[diff]):
            ta = _single_t(o, r, a); tb = _single_t(o, r, b)
            assert ta == tb
    assert (tri_b >= 0).sum() > 0.15 * W * H


def _single_t(o, ray, tri):
    """t of one ray against one triangle, via a one-triangle oracle scene."""
    one = {'v': o.v[tri:tri + 1], 'n': o.nrm[tri:tri + 1], 'mat_idx': np.zeros(1, np.uint32), 'materials': o.mats[:1]}
    s = oracle_py.Oracle(one, build_tree=False)
    return float(s.trace_rays(ray[None, :], kind=0, use_tree=False)[1][0])


@pytest.mark.parametrize('name', NAMES)
def test_front_face_by_determinant_equals_plane_normal_form(name, flats):
    """scene.d:704-705 tests plane.normal . dir < 0; the oracle uses the sign of the Moller-Trumbore determinant."""
    W = H = 64
    o = make_oracle(flats[name], name, W, H)
    assert o.frontface_disagreements(0, W * H) == 0


def test_light_table_known_answers(flats):
    """rendering.d:116-161 on the cornell box: one group (material 'Light'), two triangles, p = 0.5 each."""
    o = oracle_py.Oracle(flats['cornell-box'], build_tree=False)
    groups = o.light_groups()
    assert len(groups) == 1
    rows, area = groups[0]
    assert rows.shape == (2, 11)
    assert abs(area - 48.8846) < 1e-3
    assert np.allclose(rows[:, 9], 24.4423, atol=1e-3) and np.allclose(rows[:, 10], 0.5)
    assert abs(rows[:, 10].sum() - 1.0) < 1e-6
    assert np.allclose(rows[:, 2], 19.921038, atol=1e-5)          # v0.z
    for n in ('teapot', 'chest1'):                                # unnamed material -> grey, not emissive
        assert oracle_py.Oracle(flats[n], build_tree=False).light_groups() == []


def test_camera_state():
    """camera.d:16-29: tan of the FULL fov angle; right = up x dir; up' = right x dir points to -up (row 0 = top)."""
    c = oracle_py.camera_lookat(30.0, 0.75, (-1, 26.5, 10), (0, 0, 9), UP)
    pos, d, r, u = c[0:3], c[3:6], c[6:9], c[9:12]
    assert abs(c[12] - np.tan(np.pi / 6)) < 1e-6 and c[13] == np.float32(0.75)
    assert np.allclose(pos, [-1, 26.5, 10])
    for a in (d, r, u):
        assert abs(np.linalg.norm(a) - 1) < 1e-6
    assert abs(d @ r) < 1e-6 and abs(d @ u) < 1e-6 and abs(r @ u) < 1e-6
    assert u[2] < 0 and r[2] == 0


def test_view_ray_layout(flats):
    """rendering.d:280-283: pixel corner coordinates, x = col/W*2-1, y = row/H*2-1; centre pixel looks along dir."""
    W, H = 64, 48
    o = make_oracle(flats['teapot'], 'teapot', W, H)
    rays = o.view_rays(0, W * H).reshape(H, W, 6)
    assert np.allclose(rays[..., :3], CAMERAS['teapot'][0])
    assert np.allclose(rays[H // 2, W // 2, 3:], o.cam[3:6], atol=1e-6)
    assert rays[0, W // 2, 5] > rays[H - 1, W // 2, 5]             # first row looks further up (+z) than the last
    assert np.allclose(np.linalg.norm(rays[..., 3:], axis=2), 1, atol=1e-6)


def test_rng_stream_is_stable():
    a = oracle_py.keyed_stream(2024, 17, 3, 1, 16)
    b = oracle_py.keyed_stream(2024, 17, 3, 1, 16)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, oracle_py.keyed_stream(2024, 17, 3, 2, 16))
    assert not np.array_equal(a, oracle_py.keyed_stream(2024, 18, 3, 1, 16))
    assert not np.array_equal(a, oracle_py.keyed_stream(2024, 17, 4, 1, 16))
    assert not np.array_equal(a, oracle_py.keyed_stream(2025, 17, 3, 1, 16))
    w = oracle_py.keyed_stream(1, 2, 3, 0, 1 << 16).astype(np.float64) / 2 ** 32
    assert abs(w.mean() - 0.5) < 0.01 and abs(w.var() - 1 / 12) < 0.005


@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_golden(name, flats):
    g = np.load(os.path.join(GOLDEN, 'oracle_%s_64x64.npz' % name))
    W = H = 64
    sky = name != 'cornell-box'
    o = make_oracle(flats[name], name, W, H, build_tree=True, has_sky=sky, seed=2024)
    tri, t, u, v = o.trace_primary(0, W * H, use_tree=False)
    assert np.array_equal(tri, g['tri'])
    for got, key in ((t, 't'), (u, 'u'), (v, 'v')):
        assert np.array_equal(got.view(np.uint32), g[key].view(np.uint32))
    assert np.array_equal(o.trace_primary(0, W * H, use_tree=True)[0], g['tri_tree'])
    px = oracle_py.new_pixels(W * H)
    o.render_keyed(px, 0, 64, nthreads=4, use_tree=False)
    assert np.allclose(px['color'], g['radiance_mode0_64spp'], rtol=1e-5, atol=1e-6)   # libm sin/cos may differ by an ulp
    assert (px['n'] == 64).all()
    assert np.array_equal(oracle_py.keyed_stream(2024, 17, 3, 1, 16), g['rng_words'])


def test_keyed_render_is_partition_and_thread_independent(flats):
    W = H = 32
    o = make_oracle(flats['cornell-box'], 'cornell-box', W, H, seed=5)
    a = oracle_py.new_pixels(W * H); o.render_keyed(a, 0, 8, nthreads=1)
    b = oracle_py.new_pixels(W * H); o.render_keyed(b, 0, 8, nthreads=7)
    assert np.array_equal(a['sum'], b['sum'])
    c = oracle_py.new_pixels(W * H)
    o.render_keyed(c[:500], 0, 8); o.render_keyed(c[500:], 500, 8)
    assert np.array_equal(a['sum'], c['sum'])
    d = oracle_py.new_pixels(W * H)                               # progressive: 3 + 5 samples == 8 samples
    o.render_keyed(d, 0, 3); o.render_keyed(d, 0, 5)
    assert np.array_equal(a['sum'], d['sum']) and (d['n'] == 8).all()


def test_reference_shaped_and_keyed_renderers_agree_statistically(flats):
    """Mt19937-per-band (main.d:117) and keyed streams estimate the same image: difference within Monte Carlo noise."""
    W = H = 48
    o = make_oracle(flats['cornell-box'], 'cornell-box', W, H, build_tree=True, seed=11)
    spp = 256
    a = oracle_py.new_pixels(W * H); o.render_mt(a, spp, nthreads=8, use_tree=True)
    b = oracle_py.new_pixels(W * H); o.render_keyed(b, 0, spp, nthreads=8, use_tree=False)
    o.configure(width=W, height=H, seed=12)
    c = oracle_py.new_pixels(W * H); o.render_keyed(c, 0, spp, nthreads=8, use_tree=False)
    rm = lambda x, y: np.sqrt(((x - y) ** 2).mean()) / y.mean()
    noise = rm(b['color'], c['color'])
    assert rm(a['color'], b['color']) < 2.0 * noise
    assert abs(a['color'].mean() - b['color'].mean()) < 0.01 * b['color'].mean()
    assert (a['n'] == spp).all()


def test_contribution_asserts(flats):
    """rendering.d:421-423: G finite and >= 0, light contributions >= 0 -> no negative or non-finite radiance
    from lights.  (The sun term can be slightly negative: it is gated by n.sampleDir but weighted by n.g_sunDir,
    rendering.d:436,442; at most sin(0.9 deg)*0.7/pi per vertex.)"""
    W = H = 32
    for name in NAMES:
        for mode in (0, 1):
            sky = name != 'cornell-box'
            o = make_oracle(flats[name], name, W, H, has_sky=sky, mode=mode, seed=3)
            px = oracle_py.new_pixels(W * H); o.render_keyed(px, 0, 4, nthreads=4)
            assert np.isfinite(px['sum']).all() and (px['sum'] >= (-0.0036 * 4 * 6 if sky else 0)).all()


def test_tonemap_matches_formula():
    """main.d:87-93: saturate, sRGB OETF, truncation to ubyte."""
    px = oracle_py.new_pixels(7)
    px['color'] = np.array([[-1, 0, 0.002], [0.0031308, 0.0031309, 0.5], [1, 2, 0.9999], [0.25, 0.75, 0.1],
                            [0.01, 0.02, 0.03], [0.2, 0.4, 0.6], [0.999, 0.0001, 0.05]], np.float32)
    got = oracle_py.tonemap(px)
    c = np.clip(px['color'].astype(np.float32), 0, 1)
    lin = c * np.float32(12.92)
    gam = np.float32(1.055) * np.power(c, np.float32(1 / 2.4), dtype=np.float32) - np.float32(0.055)
    exp = np.where(c <= np.float32(0.0031308), lin, gam)
    assert np.abs(got.astype(int) - (exp * 255).astype(np.uint8).astype(int)).max() <= 1
    assert tuple(got[0]) == (0, 0, int(0.002 * 12.92 * 255))
    # 1.055f * 1 - 0.055f is 0.99999994f, and the cast truncates: a saturated channel is 254 on the scalar path
    assert tuple(got[2][:2]) == (254, 254)


def test_oracle_is_frozen():
    """Digests of the oracle's results on fixed inputs, one group per chosen thBase semantic (oracle.cpp S1..S14),
    against tests/golden/oracle_freeze.json.  Nothing in the reference pins these results, so this is what keeps a
    later change of a chosen semantic from moving the oracle and the product together unnoticed."""
    import json
    import oracle_digests
    want = json.load(open(os.path.join(GOLDEN, 'oracle_freeze.json')))
    got = oracle_digests.digests()
    assert sorted(got) == sorted(want)
    changed = [k for k in want if got[k] != want[k]]
    assert not changed, 'oracle results changed: %s' % changed


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_fixed_sincos_and_pow_against_libm():
    """S13 / S14: the fixed evaluations of sinf, cosf (binary32) and powf(x, 1/2.4) (binary64, rounded once) stay within
    1 ulp of the correctly rounded value (and of this host's libm), on the ranges the integrator and the tone map use.
    (scripts/probes/sincos_f32_check.cpp walks every binary32 in a range; on [0, 100] the bound is 1 ulp as well.)"""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 400000), rng.uniform(0, np.pi / 2, 100000),
                        [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, np.pi / 2 * 0.99]]).astype(np.float32)
    s, c = oracle_py.sincos(x)
    exact_s = np.sin(x.astype(np.float64)).astype(np.float32); exact_c = np.cos(x.astype(np.float64)).astype(np.float32)
    assert _ulps(s, exact_s).max() <= 1 and _ulps(c, exact_c).max() <= 1
    assert (s != exact_s).mean() < 0.25 and (c != exact_c).mean() < 0.25          # the correctly rounded value for > 4 in 5
    assert _ulps(s, np.sin(x)).max() <= 1 and _ulps(c, np.cos(x)).max() <= 1        # libm's float32 routines
    y = np.concatenate([rng.uniform(0.0031308, 1.0, 400000), [0.0031309, 0.25, 0.5, 1.0]]).astype(np.float32)
    p = oracle_py.pow_inv24(y)
    exact_p = np.power(y.astype(np.float64), 1 / 2.4).astype(np.float32)
    assert _ulps(p, exact_p).max() <= 1 and (p != exact_p).mean() < 1e-4
    assert p[-1] == 1.0


def test_fixed_sincos_against_libm_over_whole_images(flats):
    """What choosing S13 (a fixed sin / cos sequence) over the host's libm changes in an image: the directions differ by
    at most an ulp, which moves a radiance value in its last bits and, rarely, flips a visibility decision of one
    sample.  Bounded here per model and mode on the same keyed streams: no difference in which rays are traced beyond
    a handful, relative RMSE of the image far below the Monte Carlo noise."""
    W = H = 48
    try:
        for name in ('cornell-box', 'teapot', 'chest1'):
            for mode, spp in ((0, 8), (1, 4)):
                imgs, counts = [], []
                for libm in (False, True):
                    oracle_py.set_libm_trig(libm)
                    o = make_oracle(flats[name], name, W, H, has_sky=name != 'cornell-box', mode=mode, seed=11)
                    px = oracle_py.new_pixels(W * H)
                    _, cn = o.render_keyed(px, 0, spp, nthreads=4, use_tree=False)
                    imgs.append(px['color'].astype(np.float64)); counts.append(np.asarray(cn[:2], np.int64))
                rmse = np.sqrt(((imgs[0] - imgs[1]) ** 2).mean()) / max(imgs[0].mean(), 1e-9)
                # (measured: <= 2e-9 everywhere except cornell mode 1, 6e-3 -- a bounce direction an ulp off moves a
                # hit point on a wall and with it the 1/d^2 of its light sample; two seeds differ by ~1 at this spp)
                assert rmse < 2e-2, (name, mode, rmse)
                assert np.abs(counts[0] - counts[1]).max() <= 1e-4 * counts[0].max() + 4, (name, mode, counts)
    finally:
        oracle_py.set_libm_trig(False)


def test_keyed_sample_stride_interleaves_ranks(flats):
    """Two renderers that each draw all pixels with sampleBase = rank, sampleStride = 2 use disjoint sample indices over
    any number of passes: together, pass by pass, they draw exactly the samples one renderer draws with stride 1."""
    W = H = 24
    o = make_oracle(flats['teapot'], 'teapot', W, H, has_sky=True, seed=3)
    one = oracle_py.new_pixels(W * H)
    per_sample = []
    for k in range(8):                                  # radiance of each single sample index 0..7
        p = oracle_py.new_pixels(W * H); o.render_keyed(p, 0, 1, sample_base=k); per_sample.append(p['sum'].copy())
    for rank in range(2):
        acc = oracle_py.new_pixels(W * H)
        for _ in range(2):                              # two passes of two samples each, n carried across passes
            o.render_keyed(acc, 0, 2, sample_base=rank, sample_stride=2)
        want = np.zeros_like(acc['sum'])
        for k in range(rank, 8, 2):
            want = want + per_sample[k]                 # same order of addition as the renderer: ascending index
        assert np.array_equal(acc['sum'], want) and (acc['n'] == 4).all()
