"""Probe: N handles (N HIP streams) render interleaved 8-row bands of the frame concurrently.  Prints ms per full
frame at 256 spp for N = 1..4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as P

W, H, SPP = 1920, 1080, 256
model = os.path.join(ROOT, 'tests', 'golden', 'models', 'teapot.thModel')


def make():
    pt = P.PathTracer(W, H, device=0, has_sky=1, mode=0, seed=0x5EED)
    pt.Scene(model)
    pt.set_camera(P.Camera(30.0, H / W).setTransform((25, 10, 20), (0, 0, 0), (0, 0, 1)))
    return pt


def run(tracers, reps=3):
    n = len(tracers)
    for spp in (8, SPP):
        t0 = time.perf_counter()
        for _ in range(reps):
            for r, pt in enumerate(tracers):
                pt.render_bands(8, r, n, spp)
            for pt in tracers:
                pt.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
    return dt


tr = []
for n in (1, 2, 3, 4):
    while len(tr) < n:
        tr.append(make())
    print('%d handle(s): %.2f ms per frame' % (n, run(tr[:n])), flush=True)
