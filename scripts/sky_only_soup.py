"""The 10 M-triangle soup without its light (sky and sun only), 3840x2160, 4 spp per pass, by default and with
PT_FLAG_NO_FUSE.  (Written to compare the traversal kernel that shades its hits, 24.7 ms per pass, with shade + the
sun / sky kernel whose lanes take the next ray, 17.0 + 1.0 ms; since then a scene in global memory takes the second
path by default and both lines show it.)
  python scripts/sky_only_soup.py [triangles]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import pathtracer_amd as P
from pathtracer_amd.synthetic import soup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fl = soup(n); fl['materials'][:, 0] = 0.0
W, H, spp = 3840, 2160, 4
imgs = []
for name, flags in (('default', 0), ('PT_FLAG_NO_FUSE', P.PT_FLAG_NO_FUSE)):
    pt = P.PathTracer(W, H, has_sky=1, mode=0, seed=0x5EED, flags=flags | P.PT_FLAG_TIME_KERNELS)
    pt.upload_scene(**fl)
    pt.set_camera(P.Camera(30.0, H / W).setTransform((250, 180, 160), (0, 0, 0), (0, 0, 1)))
    pt.render_frame(spp); torch.cuda.synchronize()
    s0 = pt.stats(); t0 = time.perf_counter()
    for _ in range(2): pt.render_frame(spp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    s1 = pt.stats()
    ks = {k: round((s1.kernelMs[i] - s0.kernelMs[i]) / 2, 2) for i, k in enumerate(P.PT_K_NAMES) if s1.kernelMs[i] - s0.kernelMs[i] > 0}
    print('%-18s %.2f ms per pass  %s' % (name, dt * 1e3, ks), flush=True)
    imgs.append(pt.read_pixels()['sum'].view(np.uint32).copy()); pt.close()
print('same bits:', np.array_equal(imgs[0], imgs[1]))
