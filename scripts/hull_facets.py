"""Which triangles of a scene are facets of its convex hull (every vertex of the scene on or behind the triangle's plane)?
A ray that starts above such a triangle and does not head back through its plane cannot hit anything.
  python scripts/hull_facets.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import pathtracer_amd as P

for name in ('chest1', 'teapot', 'cornell-box'):
    pt = P.PathTracer(16, 16, device=-1)
    pt.Scene('tests/golden/models/%s.thModel' % name)
    sc = pt.get_scene()
    v = np.asarray(sc['v'], np.float64).reshape(-1, 3, 3)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ln = np.linalg.norm(n, axis=1); ok = ln > 0
    n[ok] /= ln[ok, None]
    allv = v.reshape(-1, 3)
    ext = np.ptp(allv, axis=0).max()
    h = (allv[None, :, :] - v[:, None, 0, :]) @ n[:, :, None]         # T x V x 1: height of every vertex above every plane
    h = h[:, :, 0]
    for tol in (0.0, 1e-6 * ext, 1e-5 * ext):
        front = ok & (h.max(axis=1) <= tol)
        back = ok & (h.min(axis=1) >= -tol)
        print('%-12s %5d triangles, extent %.2f, tol %.1e: hull facets (outward normal) %d, (inward normal) %d' % (name, len(v), ext, tol, front.sum(), back.sum()))
    pt.close()
