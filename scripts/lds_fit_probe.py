"""Render time of a scene whose LDS footprint is near an occupancy threshold, with an explicit leaf bound (the tree
is built as is) and with the library's choice (fuller leaves when that keeps one more workgroup per CU).  Usage: python scripts/lds_fit_probe.py <triangles>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as P
from pathtracer_amd.synthetic import soup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
fl = soup(n, seed=4, size=12.0)
W, H, spp = 1920, 1080, 64
for cost, leaf in ((1.0, 4), (1.0, 0)):                 # explicit leaf bound (no fitting) against the library's choice
    pt = P.PathTracer(W, H, has_sky=1, seed=1, max_leaf_tris=leaf)
    pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials'])
    pt.set_camera(P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1)))
    st = pt.stats()
    lds = (2 * st.numNodes + 3 * (n + 2)) * 16
    pt.render_frame(8); pt.synchronize()
    t0 = time.perf_counter(); pt.render_frame(spp); pt.synchronize(); dt = time.perf_counter() - t0
    print('leaf %d tris %d cost %.2f nodes %d lds %.1f KB inLds %d  %.2f ms' % (leaf, n, cost, st.numNodes, lds / 1024, st.sceneInLds, dt * 1e3), flush=True)
    pt.close()
