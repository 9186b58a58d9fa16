"""Ad-hoc GPU bring-up check: product (HIP) vs oracle on small cases, with timings."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import thmodel_ref as T, oracle_py as O
import pathtracer_amd as P

CAMS = {'cornell-box': ((-1, 26.5, 10), (0, 0, 9)), 'teapot': ((25, 10, 20), (0, 0, 0)), 'chest1': ((3, 3, 3), (0, 0, 0))}

def run(name, W, H, sky, mode, spp):
    path = os.path.join(ROOT, 'tests/golden/models', name + '.thModel')
    fl = T.load_flat(path)
    o = O.Oracle(fl, build_tree=True)
    o.configure(width=W, height=H, has_sky=sky, mode=mode, seed=7)
    cam = O.camera_lookat(30.0, H / W, CAMS[name][0], CAMS[name][1], (0, 0, 1))
    o.set_camera(cam)
    pt = P.PathTracer(W, H, device=0, has_sky=int(sky), mode=mode, seed=7)
    pt.Scene(path); pt.set_camera(P.Camera(30.0, H / W).setTransform(CAMS[name][0], CAMS[name][1], (0, 0, 1)))
    st = pt.stats(); print(name, 'T', st.numTriangles, 'nodes', st.numNodes, 'depth', st.bvhDepth, 'lds', st.sceneInLds)
    # rng + rays
    assert np.array_equal(pt.rng_stream(5, 3, 2, 64), O.keyed_stream(7, 5, 3, 2, 64)), 'rng'
    assert np.array_equal(pt.view_rays().view(np.uint32), o.view_rays(0, W * H).view(np.uint32)), 'view rays'
    t0 = time.time(); tri, t, u, v = pt.trace_primary(); t1 = time.time()
    otri, ot, ou, ov = o.trace_primary(0, W * H, False)
    print('  primary: gpu %.3fs  triId mismatches %d / %d  hitfrac %.3f' % (t1 - t0, (tri != otri).sum(), W * H, (otri >= 0).mean()))
    hit = otri >= 0
    same = hit & (tri == otri)
    print('  t bit-equal %s  u %s  v %s' % (np.array_equal(t[same].view(np.uint32), ot[same].view(np.uint32)),
          np.array_equal(u[same].view(np.uint32), ou[same].view(np.uint32)), np.array_equal(v[same].view(np.uint32), ov[same].view(np.uint32))))
    # image
    px = O.new_pixels(W * H)
    t0 = time.time(); sec, cn = o.render_keyed(px, 0, spp, nthreads=16, use_tree=False); t1 = time.time()
    pt.reset_stats()
    t2 = time.time(); pt.render_frame(spp); pt.synchronize(); t3 = time.time()
    g = pt.read_pixels()
    s = pt.stats()
    rays_gpu = s.raysPrimary + s.raysLight + s.raysIndirect + s.raysOcclusion
    print('  oracle %.2fs rays %d+%d | gpu %.4fs rays %d (p %d l %d i %d o %d) renderMs %.2f' % (t1 - t0, cn[0], cn[1], t3 - t2, rays_gpu, s.raysPrimary, s.raysLight, s.raysIndirect, s.raysOcclusion, s.renderMs))
    a, b = px['color'].astype(np.float64), g['color'].astype(np.float64)
    d = np.abs(a - b)
    print('  image: mean ref %.5f gpu %.5f  max abs diff %.3g  pixels differing > 1e-4: %d  relRMSE %.3g  n ok %s' % (
        a.mean(), b.mean(), np.nanmax(d), (d.max(1) > 1e-4).sum(), np.sqrt(np.nanmean(d ** 2)) / a.mean(), np.array_equal(px['n'], g['n'])))
    print('  Mrays/s gpu %.1f' % (rays_gpu / (s.renderMs * 1e-3) / 1e6))

if __name__ == '__main__':
    run('cornell-box', 256, 256, False, 0, 8)
    run('teapot', 320, 240, True, 0, 8)
    run('chest1', 256, 256, True, 1, 4)
    run('cornell-box', 128, 128, False, 1, 4)
