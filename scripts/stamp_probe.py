"""Diagnostic (with a library built -DPT_STAMP): share of the traversal's wave time spent in the descent loop and in the
leaf phase, per active lane, for the any-hit and closest-hit kernels of a workload.  In that build the visit counters
hold clock ticks summed over LANES (every active lane adds the wave's ticks), so ticks / rays = wave ticks per ray
weighted by the lanes that were active in that stretch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as P
wl = dict(model='teapot', W=1920, H=1080, frm=(25, 10, 20), to=(0, 0, 0))
pt = P.PathTracer(wl['W'], wl['H'], device=0, has_sky=1, seed=0x5EED, flags=P.PT_FLAG_COUNT_VISITS)
pt.Scene(os.path.join('tests', 'golden', 'models', wl['model'] + '.thModel'))
pt.set_camera(P.Camera(30.0, wl['H'] / wl['W']).setTransform(wl['frm'], wl['to'], (0, 0, 1)))
pt.render_frame(16); pt.synchronize()
st = pt.stats()
nc = st.raysPrimary - st.raysCulled
print('closest: rays %d  descent ticks/ray %.1f  leaf ticks/ray %.1f' % (nc, st.nodesClosest / nc, st.trisClosest / nc))
if os.environ.get('PT_SLOTS'):
    # steps and waiting slots share one 32-bit counter per lane, flushed per wave: only their sum survives the carry, so
    # the probe renders few samples per lane; report raw sums
    pass
print('any    : rays %d  descent ticks/ray %.1f  leaf ticks/ray %.1f' % (st.raysOcclusion, st.nodesAny / st.raysOcclusion, st.trisAny / st.raysOcclusion))
