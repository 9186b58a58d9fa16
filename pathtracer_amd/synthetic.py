"""Synthetic scenes for stress runs: the triangle soup of the 10 M-triangle configuration."""
import numpy as np


def soup(n_tris, seed=0x5EED, extent=100.0, size=0.5):
    """Triangle soup: centres uniform in [-extent, extent]^3, three offsets uniform in [-size, size]^3, geometric
    normals, one grey material plus an emissive quad above.  Returns the arrays PathTracer.upload_scene takes."""
    rng = np.random.default_rng(seed)
    v = np.empty((n_tris + 2, 3, 3), np.float32)
    step = 1 << 20                                     # in slices: keeps the float64 temporaries small at 10 M triangles
    for a in range(0, n_tris, step):
        b = min(n_tris, a + step)
        c = rng.uniform(-extent, extent, (b - a, 1, 3)).astype(np.float32)
        v[a:b] = (c + rng.uniform(-size, size, (b - a, 3, 3)).astype(np.float32)).astype(np.float32)
    q = extent * 0.1
    z = np.float32(extent * 1.1)
    v[n_tris:] = np.array([[[-q, -q, z], [-q, q, z], [q, q, z]], [[-q, -q, z], [q, q, z], [q, -q, z]]], np.float32)  # faces -z
    e1 = v[:, 1] - v[:, 0]; e2 = v[:, 2] - v[:, 0]
    n = np.cross(e1, e2)
    ln = np.linalg.norm(n, axis=1, keepdims=True); ln[ln == 0] = 1
    n = (n / ln).astype(np.float32)
    nrm = np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)
    mat_idx = np.zeros(len(v), np.uint32); mat_idx[-2:] = 1
    mats = np.array([[0.0, 0.7, 0.7, 0.7], [0.5, 0.7, 0.7, 0.7]], np.float32)
    return {'v': v, 'n': nrm, 'mat_idx': mat_idx, 'materials': mats}
