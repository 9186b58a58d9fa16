"""The operators on the 10 M-triangle soup (scene in global memory): Scene.trace of the 3840x2160 view rays
(trace_primary), Scene.trace and Scene.hitsNothing of 4 M rays between random points of the scene's box.  Prints wall
times (copies included); run it under rocprofv3 --kernel-trace --stats for the kernels' own times (scripts/trace_py.sh)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import pathtracer_amd as P
from pathtracer_amd.synthetic import soup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W, H = 3840, 2160
pt = P.PathTracer(W, H, has_sky=1, mode=0, seed=0x5EED)
pt.upload_scene(**soup(n))
pt.set_camera(P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1)))
rng = np.random.default_rng(7)
a = rng.uniform(-100, 100, (4_000_000, 3)).astype(np.float32); b = rng.uniform(-100, 100, (4_000_000, 3)).astype(np.float32)
d = b - a; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([a, d], axis=1).astype(np.float32)
for name, f in (('trace_primary', lambda: pt.trace_primary()), ('trace', lambda: pt.trace(rays)), ('hitsNothing', lambda: pt.hitsNothing(rays))):
    f(); t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    x = r[0] if isinstance(r, tuple) else r
    print('%-14s %.1f ms  checksum %d' % (name, dt * 1e3, int(np.asarray(x, np.int64).sum())), flush=True)
pt.close()
