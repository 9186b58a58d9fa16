"""Differential fuzz of Scene.trace / Scene.hitsNothing on the GPU against the exhaustive oracle: random soups of
random scale, offset and triangle size (with degenerate and duplicated triangles mixed in), random rays, scene in LDS
and in global memory.  Prints one line per case; exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_py
import pathtracer_amd as P
from pathtracer_amd.synthetic import soup
from test_gpu_parity import random_rays

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2024)
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 7, 40, 300, 1200, 3000]))
    extent = float(10.0 ** rng.uniform(-2, 4)); size = extent * float(10.0 ** rng.uniform(-3, 0)); offset = extent * float(rng.choice([0, 0, 10, -300]))
    fl = soup(n, seed=int(rng.integers(1 << 30)), extent=extent, size=size)
    v = fl['v'] + np.float32(offset)
    if n >= 7:
        v[3] = v[3][[0, 0, 1]]                      # a degenerate triangle
        v[5] = v[4]                                  # a duplicate: equal t, the lower id must win
    fl['v'] = v.astype(np.float32)
    eps = 1e-5 * max(1.0, extent / 100.0) if extent >= 1 else 1e-9
    o = oracle_py.Oracle(fl, build_tree=False); o.configure(width=8, height=8, eps=eps)
    rays = random_rays(fl, 8000, int(rng.integers(1 << 30)))
    ref = o.trace_rays(rays, kind=0, use_tree=False); free_ref = o.trace_rays(rays, kind=1, use_tree=False)[0] < 0
    for flags in (0, P.PT_FLAG_NO_LDS_SCENE):
        pt = P.PathTracer(8, 8, flags=flags, eps=eps)
        pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
        tri, t, u, vv, _ = pt.trace(rays)
        ok = np.array_equal(tri, ref[0]) and np.array_equal(t[ref[0] >= 0].view(np.uint32), ref[1][ref[0] >= 0].view(np.uint32))
        ok = ok and np.array_equal(pt.hitsNothing(rays), free_ref)
        print('case %2d tris %4d extent %.3g size %.3g offset %.3g lds %d hits %d: %s' %
              (c, n, extent, size, offset, pt.stats().sceneInLds, int((ref[0] >= 0).sum()), 'ok' if ok else 'MISMATCH'), flush=True)
        pt.close()
        if not ok:
            bad = np.flatnonzero(tri != ref[0])[:5]
            print('first differing rays', bad, tri[bad], ref[0][bad]); sys.exit(1)
print('all cases equal')
