"""Differential fuzz of the work-saving render paths against the general one: random soups from a few triangles (the
shading kernel traces its own rays, whole-sample kernel) to a few thousand (raygen inside extend, the traversal kernel
that shades its hits) and scenes forced into global memory, random emissive materials, sky on or off, both modes,
windows that are not whole rows.  Every accumulator and counter must equal what PT_FLAG_NO_FUSE renders, and the
accumulators the oracle's image with the same streams, bit for bit, in both modes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import oracle_py
import pathtracer_amd as P
from pathtracer_amd.synthetic import soup

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(77)
FIELDS = ('raysPrimary', 'raysLight', 'raysIndirect', 'raysOcclusion', 'raysCulled', 'hits', 'hitsPrimary', 'envEntries', 'samples')
for c in range(cases):
    n = int(rng.choice([1, 3, 12, 40, 90, 400, 1500, 3000]))
    size = float(rng.choice([4.0, 12.0, 30.0]))
    fl = soup(n, seed=int(rng.integers(1 << 30)), size=size)
    if rng.random() < 0.5:                                # no emissive material: sun / sky only
        fl['materials'][:, 0] = 0.0
    sky = int(rng.random() < 0.7 or not (fl['materials'][:, 0] > 0).any())
    mode = int(rng.random() < 0.5)
    glob = P.PT_FLAG_NO_LDS_SCENE if rng.random() < 0.25 else 0
    if glob and c % 2 == 0: glob |= P.PT_FLAG_HIT_ORDER       # (default only for geometry as fine as the pixels)
    W, H = int(rng.choice([64, 96, 120])), int(rng.choice([48, 64]))
    cam = P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1))
    out = []
    depth, chains = int(rng.choice([1, 3, 5])), int(rng.choice([1, 2, 5]))
    for extra in (0, P.PT_FLAG_QUEUED_VIEW_RAYS, P.PT_FLAG_NO_PIXEL_SUMS, P.PT_FLAG_NO_FUSE):
        pt = P.PathTracer(W, H, has_sky=sky, mode=mode, seed=c, max_depth=depth, n_indirect=chains, flags=glob | extra)
        pt.upload_scene(fl['v'], fl['n'], fl['mat_idx'], fl['materials']); pt.set_camera(cam)
        pt.render_frame(3)
        pt.render(7, W * H - 13, 2)                       # a window that is not whole rows
        out.append((pt.read_pixels(), pt.stats()))
        pt.close()
    ok = True
    for px, st in out[:3]:
        ok = ok and np.array_equal(px['n'], out[3][0]['n']) and np.array_equal(px['sum'].view(np.uint32), out[3][0]['sum'].view(np.uint32))
        ok = ok and all(getattr(st, f) == getattr(out[3][1], f) for f in FIELDS)
    o = oracle_py.Oracle(fl, build_tree=False)
    o.configure(width=W, height=H, has_sky=bool(sky), mode=mode, seed=c, max_depth=depth, n_indirect=chains); o.set_camera(cam.state)
    ref = oracle_py.new_pixels(W * H)
    o.render_keyed(ref, 0, 3, nthreads=8)
    win = ref[7:7 + W * H - 13]; o.render_keyed(win, 7, 2, nthreads=8); ref[7:7 + W * H - 13] = win
    same = np.array_equal(out[0][0]['n'], ref['n']) and np.array_equal(out[0][0]['sum'].view(np.uint32), ref['sum'].view(np.uint32))
    ok = ok and same
    st = out[0][1]
    print('case %2d tris %4d size %4.0f lights %d sky %d mode %d global %d  in-shade rays p/l/i/o %d/%d/%d/%d of %d/%d/%d/%d: %s' %
          (c, n, size, st.numLightGroups, sky, mode, 1 if glob else 0, st.raysPrimaryInShade, st.raysLightInShade, st.raysIndirectInShade,
           st.raysOcclusionInShade, st.raysPrimary - st.raysCulled, st.raysLight, st.raysIndirect, st.raysOcclusion,
           ('ok' if ok else 'MISMATCH') + ('' if same else ' (oracle)')), flush=True)
    if not ok:
        sys.exit(1)
print('all cases equal')
