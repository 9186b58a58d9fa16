"""Wall time of mode-1 frames (five indirect chains) of the small scenes: scripts/time_mode1.py [spp]
(PTGPU_LIBRARY picks the library build, as for bench.py)."""
import sys, time
sys.path.insert(0, 'tests')
from conftest import make_tracer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, sky in (('cornell-box', 0), ('cornell-box', 1), ('chest1', 1)):
    pt = make_tracer(name, 1920, 1080, has_sky=sky, mode=1, seed=3)
    pt.render_frame(spp); pt.read_pixels()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); pt.render_frame(spp); pt.synchronize(); best = min(best, time.perf_counter() - t0)
    st = pt.stats()
    print('%s sky=%d mode 1, 1920x1080, %d spp: %.2f ms  (indirect rays %.0f M, hits %.0f M)' %
          (name, sky, spp, best * 1e3, st.raysIndirect / 1e6 / 4, st.hits / 1e6 / 4))
    pt.close()
