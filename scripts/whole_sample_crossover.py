"""Times one batch of the traced-shading path with and without the whole-sample kernel over batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pathtracer_amd as P
from conftest import make_tracer

for name, sky, mode in (('cornell-box', 0, 0), ('chest1', 1, 1), ('chest1', 1, 0)):
    for W, H, spp in ((512, 512, 16), (1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 8), (1920, 1080, 16), (1920, 1080, 32)):
        row = []
        for flags in (P.PT_FLAG_QUEUED_VIEW_RAYS, 0):
            pt = make_tracer(name, W, H, has_sky=sky, mode=mode, flags=flags | P.PT_FLAG_NO_GRAPH)
            pt.render_frame(spp); pt.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                pt.render_frame(spp)
            pt.synchronize()
            row.append((time.perf_counter() - t0) / 5 * 1e3)
            pt.close()
        print('%-12s mode %d %4dx%-4d spp %2d  %6.1f M slots  separate %7.3f ms  whole-sample %7.3f ms' %
              (name, mode, W, H, spp, W * H * spp / 1e6, row[0], row[1]), flush=True)
