"""What the parts of a chest1 1080p sample cost: one 64-spp batch with and without the sky (sun and sky rays of a
vertex), with and without the chains (mode), recorded and summed in registers.  scripts/time_chest_parts.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import pathtracer_amd as P
from conftest import make_tracer

for label, sky, mode, flags in (('mode 1, sky', 1, 1, 0), ('mode 1, no sky', 0, 1, 0), ('mode 0, sky, recorded', 1, 0, P.PT_FLAG_NO_PIXEL_SUMS),
                                ('mode 0, sky, summed in registers', 1, 0, 0), ('mode 0, no sky, recorded', 0, 0, P.PT_FLAG_NO_PIXEL_SUMS)):
    pt = make_tracer('chest1', 1920, 1080, has_sky=sky, mode=mode, seed=3, flags=flags | P.PT_FLAG_TIME_KERNELS)
    pt.render_frame(64); pt.synchronize(); pt.reset_stats()
    for _ in range(3):
        pt.render_frame(64)
    pt.synchronize()
    st = pt.stats()
    k = {n: (l, ms / max(l, 1) * 1e3) for n, l, ms in zip(P.PT_K_NAMES, st.kernelLaunches, st.kernelMs) if l}
    print('%-34s' % label, ' '.join('%s %d x %.0f us' % (n, l, us) for n, (l, us) in k.items()))
    pt.close()
