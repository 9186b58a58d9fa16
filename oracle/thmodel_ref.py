"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Independent numpy restatement of two host-side steps of the reference so that the
product's C++ reader/flattener has something to be compared with:

  * the .thModel chunk-file read sequence      -> /root/reference/modelloader.d:125-704
  * Scene's flattening of the node graph       -> /root/reference/scene.d:209-321
  * the name -> Material table                  -> /root/reference/rendering.d:247-275
  * the per-vertex flag bits                    -> /root/reference/modeltypes.d:28-38

Parity status: the reference has no tests; the only known answers are properties
of the three shipped model files (triangle counts, bounding boxes, areas).  Those
are pinned in tests/test_thmodel.py.  thBase (the reference's math library) is not
part of the reference checkout, so the mat4 convention below is a *chosen*
semantic (row vectors, row-major 16 floats):  v' = v * M.
"""
import struct
import numpy as np

F32 = np.float32

# modeltypes.d:28-38
PVD_POSITION, PVD_NORMAL, PVD_TANGENT, PVD_BITANGENT = 1, 2, 4, 8
PVD_TEXCOORD0 = 0x10


class _Cur:
    def __init__(self, data, pos=0, end=None):
        self.d, self.p, self.end = data, pos, len(data) if end is None else end

    def u8(self):
        if self.p + 1 > len(self.d):
            raise ValueError('truncated file')
        v = self.d[self.p]; self.p += 1; return v

    def u32(self):
        if self.p + 4 > len(self.d):
            raise ValueError('truncated file')
        v = struct.unpack_from('<I', self.d, self.p)[0]; self.p += 4; return v

    def raw(self, n):
        v = self.d[self.p:self.p + n]; self.p += n
        if len(v) != n:
            raise ValueError('truncated file')
        return v

    def arr(self, dtype, n):
        a = np.frombuffer(self.raw(np.dtype(dtype).itemsize * n), dtype=dtype).copy()
        return a

    def chunk(self, expect=None):
        """Chunk := u8 nameLen, name, u32 payloadLen, payload (framing inferred from the files)."""
        n = self.u8(); name = self.raw(n).decode('latin1'); ln = self.u32()
        if expect is not None and name != expect:
            raise ValueError("Expected '%s' chunk but got '%s'" % (expect, name))
        c = _Cur(self.d, self.p, self.p + ln); c.name = name
        self.p += ln
        return c

    def peek_chunk_name(self):
        n = self.d[self.p]
        return self.d[self.p + 1:self.p + 1 + n].decode('latin1')

    def done(self):
        return self.p == self.end


def read_thmodel(path):
    """Follows modelloader.d:137-703 (LoadFile with Load.Everything)."""
    data = open(path, 'rb').read()
    top = _Cur(data)
    f = top.chunk('thModel')                               # modelloader.d:139
    if f.end != len(data):
        raise ValueError('trailing bytes after thModel chunk')
    version = f.u32(); f.u32()                             # version, debug flag
    if version > 2:                                        # modelloader.d:144
        raise ValueError('unsupported version %d' % version)
    model = {'version': version}

    s = f.chunk('sizeinfo')                                # modelloader.d:196-327
    s.u32(); s.u32()
    if version >= 2:
        s.u32()
    s.u32(); nmesh = s.u32()
    size_meshes = []
    for _ in range(nmesh):
        nv = s.u32(); flags = s.u32()
        ntc = bin(flags & 0xF0).count('1')
        comps = [s.u8() for _ in range(ntc)]
        nf = s.u32()
        size_meshes.append((nv, flags, comps, nf))
    for _ in range(5):
        s.u32()
    assert s.done()
    model['sizeinfo_meshes'] = size_meshes

    t = f.chunk('textures')                                # modelloader.d:338-374
    model['textures'] = [t.raw(t.u32()).decode('latin1') for _ in range(t.u32())]
    assert t.done()

    m = f.chunk('materials')                               # modelloader.d:377-445
    mats = []
    for _ in range(m.u32()):
        c = m.chunk('mat')
        name = c.raw(c.u32()).decode('latin1') if version >= 2 else ''
        texs = [(c.u32(), c.u8()) for _ in range(c.u32())]
        assert c.done()
        mats.append({'name': name, 'textures': texs})
    assert m.done()
    model['materials'] = mats

    ms = f.chunk('meshes')                                 # modelloader.d:448-633
    meshes = []
    for _ in range(ms.u32()):
        c = ms.chunk('mesh')
        mesh = {'materialIndex': c.u32(), 'bbox_min': c.arr('<f4', 3), 'bbox_max': c.arr('<f4', 3)}
        nv = c.u32()
        v = c.chunk('vertices')
        mesh['vertices'] = v.arr('<f4', nv * 3).reshape(nv, 3)
        assert v.done()
        mesh['normals'] = None; mesh['texcoords'] = []
        while True:
            sub = c.chunk()
            if sub.name in ('normals', 'tangents', 'bitangents'):
                raw = sub.arr('<i2', nv * 3).reshape(nv, 3)
                # UncompressFloat, modelloader.d:492-497: cast(float)data / cast(float)short.max
                val = raw.astype(F32) / F32(32767.0)
                if sub.name == 'normals':
                    mesh['normals'] = val
                assert sub.done()
            elif sub.name == 'texcoords':
                for _k in range(sub.u8()):
                    comps = sub.u8()
                    if comps != 2:                          # modelloader.d:572
                        raise ValueError('only 2 component texture coordinates are supported')
                    mesh['texcoords'].append(sub.arr('<f4', nv * 2).reshape(nv, 2))
                assert sub.done()
            elif sub.name == 'faces':
                nf = sub.u32()
                idx = sub.arr('<u4' if nv > 65535 else '<u2', nf * 3)   # modelloader.d:602-615
                mesh['faces'] = idx.astype(np.uint32).reshape(nf, 3)
                assert sub.done()
                break
            else:
                raise ValueError("Unexpected chunk '%s'" % sub.name)
        assert c.done()
        meshes.append(mesh)
    assert ms.done()
    model['meshes'] = meshes

    nd = f.chunk('nodes')                                  # modelloader.d:636-692
    nodes = []
    for _ in range(nd.u32()):
        name = nd.raw(nd.u32()).decode('latin1')
        tr = nd.arr('<f4', 16)
        parent = nd.u32()
        mesh_idx = [nd.u32() for _ in range(nd.u32())]
        children = [nd.u32() for _ in range(nd.u32())]
        nodes.append({'name': name, 'transform': tr, 'parent': None if parent == 0xFFFFFFFF else parent,
                      'meshes': mesh_idx, 'children': children})
    assert nd.done() and f.done()
    model['nodes'] = nodes
    return model


# ---------------------------------------------------------------- materials
def fill_material(name):
    """rendering.d:247-275 -> (emissive, r, g, b)."""
    e, c = 0.0, [0.7, 0.7, 0.7]
    if name == 'Light':
        e = 0.5
    elif name == 'Red':
        c = [0.7, 0.0, 0.0]
    elif name == 'Green':
        c = [0.0, 0.7, 0.0]
    elif name == 'Blue':
        c = [0.0, 0.0, 0.7]
    return np.array([e] + c, dtype=F32)


# ---------------------------------------------------------------- mat4 helpers (chosen: row vectors)
def mat4_mul(a, b):
    """c = a*b for row-major 4x4 float32, each element ((a0*b0+a1*b1)+a2*b2)+a3*b3, unfused."""
    a = a.reshape(4, 4); b = b.reshape(4, 4)
    c = np.zeros((4, 4), F32)
    for i in range(4):
        for j in range(4):
            acc = F32(a[i, 0] * b[0, j])
            for k in (1, 2, 3):
                acc = F32(acc + F32(a[i, k] * b[k, j]))
            c[i, j] = acc
    return c.reshape(16)


def mat4_inverse_rigid_or_general(m):
    """Inverse used only to build modelMatrix = inverse(root.transform) (float64 then rounded)."""
    return np.linalg.inv(m.reshape(4, 4).astype(np.float64)).astype(F32).reshape(16)


def transform_position(m, v):
    """v' = v*M, row vectors: x' = ((x*m0 + y*m4) + z*m8) + m12, unfused float32."""
    m = m.astype(F32); v = v.astype(F32)
    out = np.empty_like(v)
    for j in range(3):
        out[:, j] = ((v[:, 0] * m[0 + j] + v[:, 1] * m[4 + j]) + v[:, 2] * m[8 + j]) + m[12 + j]
    return out


def normal_matrix(m):
    """(M3^-1)^T = cofactor(M3)/det(M3) of the upper-left 3x3, unfused float32 (row-vector use: n' = n*N)."""
    a = m.reshape(4, 4)[:3, :3].astype(F32)
    c = np.empty((3, 3), F32)
    for i in range(3):
        for j in range(3):
            i1, i2 = (i + 1) % 3, (i + 2) % 3
            j1, j2 = (j + 1) % 3, (j + 2) % 3
            c[i, j] = F32(F32(a[i1, j1] * a[i2, j2]) - F32(a[i1, j2] * a[i2, j1]))
    det = F32(F32(F32(a[0, 0] * c[0, 0]) + F32(a[0, 1] * c[0, 1])) + F32(a[0, 2] * c[0, 2]))
    return (c / det).astype(F32)


def dot3(a, b):
    """fma-shaped dot evaluated in float64 then rounded is NOT bit-equal to fmaf chains, so
    do the chain exactly: fma(az,bz, fma(ay,by, ax*bx)) using float64 products (exact for f32 inputs)."""
    ax, ay, az = (a[..., i].astype(np.float64) for i in range(3))
    bx, by, bz = (b[..., i].astype(np.float64) for i in range(3))
    t = (ax * bx).astype(F32)                               # single rounding of exact product
    t = (ay * by + t.astype(np.float64)).astype(F32)        # fma: exact product + addend, one rounding*
    t = (az * bz + t.astype(np.float64)).astype(F32)
    return t
# (*) a float64 sum of an exact 48-bit product and a float32 can itself round before the final
# float32 rounding (double rounding).  Only used for normalisation of normals, which the tests
# compare to the product with a 1-ulp tolerance, never bit-exactly.


def normalize_rows(v):
    ln = np.sqrt(dot3(v, v)).astype(F32)
    inv = (F32(1.0) / ln).astype(F32)
    return (v * inv[:, None]).astype(F32)


def flatten(model, model_matrix=None, mat_func=fill_material):
    """scene.d:209-321.  Returns dict(v[T,3,3], n[T,3,3], mat_idx[T], materials[M,4], names)."""
    nodes = model['nodes']
    root = nodes[0]
    if model_matrix is None:
        model_matrix = np.eye(4, dtype=F32).reshape(16)
    leafs = []

    def find_leafs(i):                                      # scene.d:213-223
        if nodes[i]['meshes']:
            leafs.append(i)
        for c in nodes[i]['children']:
            find_leafs(c)
    find_leafs(0)
    if not leafs:
        raise ValueError('no leaf nodes found')
    mats = np.stack([mat_func(m['name']) for m in model['materials']])   # scene.d:239-245
    V, N, MI, TX = [], [], [], []
    for leaf in leafs:
        tr = mat4_mul(root['transform'], model_matrix)      # scene.d:257
        cur = leaf
        while cur is not None and cur != 0:                 # scene.d:258-262
            tr = mat4_mul(nodes[cur]['transform'], tr)
            cur = nodes[cur]['parent']
        nm = normal_matrix(tr)                              # scene.d:264
        for mi in nodes[leaf]['meshes']:
            mesh = model['meshes'][mi]
            vt = transform_position(tr, mesh['vertices'])   # scene.d:280
            nr = mesh['normals'].astype(F32)
            nt = np.empty_like(nr)
            for j in range(3):                              # n' = n*N (row vector), unfused
                nt[:, j] = (nr[:, 0] * nm[0, j] + nr[:, 1] * nm[1, j]) + nr[:, 2] * nm[2, j]
            nt = normalize_rows(nt)                         # scene.d:288
            f = mesh['faces']
            V.append(vt[f]); N.append(nt[f])
            MI.append(np.full(len(f), mesh['materialIndex'], np.uint32))
            # scene.d:305-307 reads texcoord set 1; the shipped models only carry set 0 (the D program would stop
            # with a range error on them), so the last set present stands in, zeros if there is none
            tcs = mesh['texcoords']
            tc = tcs[1] if len(tcs) > 1 else (tcs[0] if tcs else np.zeros((len(mesh['vertices']), 2), F32))
            TX.append(tc.astype(F32)[f])
    return {'v': np.concatenate(V).astype(F32), 'n': np.concatenate(N).astype(F32),
            'mat_idx': np.concatenate(MI), 'materials': mats.astype(F32),
            'names': [m['name'] for m in model['materials']], 'tex': np.concatenate(TX).astype(F32)}


def load_flat(path, raw_mesh_space=True):
    """Benchmark world = raw mesh space with non-root node transforms applied:
    modelMatrix = inverse(root.transform) (the Scene ctor parameter, scene.d:193)."""
    model = read_thmodel(path)
    mm = mat4_inverse_rigid_or_general(model['nodes'][0]['transform']) if raw_mesh_space else None
    return flatten(model, mm)


def tri_area(v):
    """Triangle.area = 0.5*|(v1-v0) x (v2-v0)| (float64, for known-answer checks only)."""
    v = v.astype(np.float64)
    return 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)
