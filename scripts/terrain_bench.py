"""A coherent mid-size scene in global memory: a height field of N x N cells (2 N^2 triangles) under sun and sky, 1920x1080.
Times a pass of 16 spp by default, with the hit order forced on and forced off (PT_FLAG_HIT_ORDER / PT_FLAG_NO_HIT_ORDER).
  python scripts/terrain_bench.py [N]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import pathtracer_amd as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 700
x = np.linspace(-100, 100, N + 1, dtype=np.float32)
X, Y = np.meshgrid(x, x, indexing='ij')
Z = (8 * np.sin(X * 0.07) * np.cos(Y * 0.05) + 3 * np.sin(X * 0.31 + Y * 0.23) + 1.0 * np.sin(X * 1.3) * np.sin(Y * 1.7)).astype(np.float32)
Pn = np.stack([X, Y, Z], axis=-1)
a, b, c, d = Pn[:-1, :-1], Pn[1:, :-1], Pn[1:, 1:], Pn[:-1, 1:]
v = np.concatenate([np.stack([a, b, c], axis=2).reshape(-1, 3, 3), np.stack([a, c, d], axis=2).reshape(-1, 3, 3)]).astype(np.float32)
n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
flip = n[:, 2] < 0
v[flip] = v[flip][:, [0, 2, 1]]; n[flip] = -n[flip]
nrm = np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)
mat_idx = np.zeros(len(v), np.uint32); mats = np.array([[0.0, 0.7, 0.7, 0.7]], np.float32)
W, H, spp = 1920, 1080, 16
imgs = []
for name, flags in (('default', 0), ('hit order on', P.PT_FLAG_HIT_ORDER), ('hit order off', P.PT_FLAG_NO_HIT_ORDER)):
    pt = P.PathTracer(W, H, has_sky=1, mode=0, seed=5, flags=flags | P.PT_FLAG_TIME_KERNELS)
    pt.upload_scene(v, nrm, mat_idx, mats)
    pt.set_camera(P.Camera(30.0, H / W).setTransform((150, 110, 90), (0, 0, 0), (0, 0, 1)))
    pt.render_frame(spp); torch.cuda.synchronize()
    s0 = pt.stats(); t0 = time.perf_counter()
    for _ in range(3): pt.render_frame(spp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    s1 = pt.stats()
    ks = {k: round((s1.kernelMs[i] - s0.kernelMs[i]) / 3, 2) for i, k in enumerate(P.PT_K_NAMES) if s1.kernelMs[i] - s0.kernelMs[i] > 0}
    print('%d triangles, %-14s %.2f ms per %d spp  %s' % (len(v), name, dt * 1e3, spp, ks), flush=True)
    imgs.append(pt.read_pixels()['sum'].view(np.uint32).copy()); pt.close()
print('same bits:', all(np.array_equal(imgs[0], x) for x in imgs[1:]))
