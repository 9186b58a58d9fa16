"""SHA-256 digests of what the CPU oracle computes for fixed inputs, one group per chosen semantic (S1..S14 in
oracle/oracle.cpp's header).  The reference pins none of these results, so the oracle could drift together with the
product unnoticed; tests/golden/oracle_freeze.json holds the digests (written by tests/golden/make_golden.py) and
tests/test_oracle.py recomputes them.  A deliberate change of a semantic has to regenerate that file, which shows
in the history."""
import hashlib

import numpy as np

import oracle_py
import thmodel_ref
from conftest import CAMERAS, UP, model_path

W = H = 48


def _h(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        m.update(str(a.dtype).encode()); m.update(str(a.shape).encode()); m.update(a.tobytes())
    return m.hexdigest()


def digests():
    out = {}
    rng = np.random.default_rng(20240607)
    ang = rng.uniform(0, 2 * np.pi, 4096).astype(np.float32)
    out['S13_sincos'] = _h(*oracle_py.sincos(ang))
    out['S14_pow_inv24'] = _h(oracle_py.pow_inv24(rng.uniform(0.0031308, 1.0, 4096).astype(np.float32)))
    nrm = rng.normal(size=(4096, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    nrm[::7] = 0                                               # rows that take angleToLocalDirection
    phi = rng.uniform(0, np.pi / 2, 4096).astype(np.float32)
    out['S8_angle_directions'] = _h(oracle_py.angle_directions(phi, ang, nrm))
    out['S10_rng_words'] = _h(oracle_py.keyed_stream(2024, 17, 3, 1, 64))
    for name in sorted(CAMERAS):
        flat = thmodel_ref.load_flat(model_path(name))
        o = oracle_py.Oracle(flat, build_tree=True)
        sky = name != 'cornell-box'
        o.configure(width=W, height=H, has_sky=sky, seed=7)
        o.set_camera(oracle_py.camera_lookat(30.0, H / W, CAMERAS[name][0], CAMERAS[name][1], UP))
        out[name + '/camera'] = _h(o.cam)                                          # camera.d:16-29, S1, S2
        out[name + '/view_rays'] = _h(o.view_rays(0, W * H))                       # rendering.d:278-285
        tri, t, u, v = o.trace_primary(0, W * H, use_tree=False)
        out[name + '/S3_S4_S6_hits_exhaustive'] = _h(tri, t, u, v)
        tri_t, t_t, u_t, v_t = o.trace_primary(0, W * H, use_tree=True)
        out[name + '/S5_S12_hits_sphere_tree'] = _h(tri_t, t_t, u_t, v_t, np.array([oracle_py.lib().orc_num_nodes(o.h)]))
        hit = np.flatnonzero(tri >= 0)[:256]
        out[name + '/S7_normals'] = _h(np.stack([o.interp_normal(tri[i], u[i], v[i]) for i in hit]))
        out[name + '/S11_lights'] = _h(*[np.concatenate([rows.ravel(), [area]]).astype(np.float32) for rows, area in o.light_groups()] or
                                       [np.zeros(0, np.float32)])
        rays = o.view_rays(0, W * H)[::5]
        out[name + '/hits_nothing'] = _h(o.trace_rays(rays, kind=1, use_tree=False)[0], o.trace_rays(rays, kind=1, use_tree=True)[0])
        px = oracle_py.new_pixels(W * H)
        _, cn = o.render_keyed(px, 0, 4, nthreads=4, use_tree=False)
        out[name + '/image_mode0'] = _h(px['sum'], px['n'], cn[:2])
        out[name + '/tonemap'] = _h(oracle_py.tonemap(px))
        o.configure(width=W, height=H, has_sky=sky, mode=1, seed=7)
        px1 = oracle_py.new_pixels(W * H)
        _, cn1 = o.render_keyed(px1, 0, 2, nthreads=4, use_tree=False)
        out[name + '/image_mode1'] = _h(px1['sum'], px1['n'], cn1[:2])
        pm = oracle_py.new_pixels(W * H)
        _, cnm = o.render_mt(pm, 2, nthreads=3, use_tree=True)                     # bands + Mt19937 + sphere tree
        out[name + '/image_reference_shaped'] = _h(pm['sum'], pm['n'], cnm)
    return out
