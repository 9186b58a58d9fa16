"""Writes the golden vectors under tests/golden/ from the CPU oracle (oracle/).

The reference pins nothing for this path (no tests, no fixtures, no runnable build), so these vectors pin the
ORACLE: any later change to oracle/ or to the chosen thBase semantics that alters a result shows up as a diff
against them.  Inputs: the three .thModel files the reference ships (copied as data to tests/golden/models/)
and the three cameras its source holds for them (rendering.d:78, 81, 94).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import oracle_py  # noqa: E402
import thmodel_ref  # noqa: E402
from conftest import CAMERAS, UP, model_path  # noqa: E402

W = H = 64


def main():
    for name in CAMERAS:
        flat = thmodel_ref.load_flat(model_path(name))
        o = oracle_py.Oracle(flat, build_tree=True)
        sky = name != 'cornell-box'
        o.configure(width=W, height=H, has_sky=sky, seed=2024)
        o.set_camera(oracle_py.camera_lookat(30.0, H / W, CAMERAS[name][0], CAMERAS[name][1], UP))
        tri, t, u, v = o.trace_primary(0, W * H, use_tree=False)            # Oracle-B: exhaustive, lowest id on ties
        tri_t, t_t, _, _ = o.trace_primary(0, W * H, use_tree=True)         # Oracle-R: sphere tree, BFS order
        px = oracle_py.new_pixels(W * H)
        o.render_keyed(px, 0, 64, nthreads=8, use_tree=False)               # mode 0 (HEAD), keyed streams, 64 spp
        out = dict(tri=tri, t=t, u=u, v=v, tri_tree=tri_t, radiance_mode0_64spp=px['color'].copy())
        o.configure(width=W, height=H, has_sky=sky, mode=1, seed=2024)
        px1 = oracle_py.new_pixels(W * H)
        o.render_keyed(px1, 0, 16, nthreads=8, use_tree=False)              # mode 1 (+5 indirect chains), 16 spp
        out['radiance_mode1_16spp'] = px1['color'].copy()
        out['rng_words'] = oracle_py.keyed_stream(2024, 17, 3, 1, 16)
        np.savez_compressed(os.path.join(HERE, 'oracle_%s_64x64.npz' % name), **out)
        print(name, 'hits', int((tri >= 0).sum()), 'tree==brute', bool(np.array_equal(tri, tri_t)),
              'mean radiance', float(px['color'].mean()), float(px1['color'].mean()))
    import json
    import oracle_digests
    with open(os.path.join(HERE, 'oracle_freeze.json'), 'w') as f:
        json.dump(oracle_digests.digests(), f, indent=1, sort_keys=True)
        f.write('\n')
    print('oracle_freeze.json written')


if __name__ == '__main__':
    main()
