"""Known answers of the three .thModel files the reference ships, for both readers: the numpy restatement
(oracle/thmodel_ref.py) and the product's C++ reader + flattening reached through the C ABI on a host-only
handle (device = -1).  modelloader.d:125-704, scene.d:209-321, rendering.d:247-275."""
import os

import numpy as np
import pytest

import pathtracer_amd as P
import thmodel_ref
from conftest import CAMERAS, model_path

KAT = {
    # name: version, triangles per mesh, vertices per mesh, material names, nodes, file bytes
    'cornell-box': (2, [46, 2, 2, 2], [102, 4, 4, 4], ['Green', 'Light', 'Red', 'Standard'], 5, 5898),
    'teapot': (1, [1024], [709], [''], 2, 33553),
    'chest1': (1, [12], [24], [''], 2, 1572),
}
AREAS = {'cornell-box': 2880.5277, 'teapot': 1288.5809, 'chest1': 24.0}
ROOT_ROT = [1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1]


@pytest.mark.parametrize('name', sorted(KAT))
def test_reader_known_answers(name):
    ver, tris, verts, mats, nodes, size = KAT[name]
    assert os.path.getsize(model_path(name)) == size
    m = thmodel_ref.read_thmodel(model_path(name))          # parses to exact EOF or raises
    assert m['version'] == ver
    assert [len(x['faces']) for x in m['meshes']] == tris
    assert [len(x['vertices']) for x in m['meshes']] == verts
    assert [x['name'] for x in m['materials']] == mats
    assert len(m['nodes']) == nodes
    for mesh, (nv, flags, comps, nf) in zip(m['meshes'], m['sizeinfo_meshes']):
        assert flags == 0x1F and comps == [2] and nv == len(mesh['vertices']) and nf == len(mesh['faces'])
        assert np.allclose(mesh['vertices'].min(0), mesh['bbox_min'], atol=1e-4)
        assert np.allclose(mesh['vertices'].max(0), mesh['bbox_max'], atol=1e-4)
    rot_node = 1 if name == 'chest1' else 0
    assert list(m['nodes'][rot_node]['transform']) == ROOT_ROT


def test_teapot_bbox_and_degenerates():
    fl = thmodel_ref.load_flat(model_path('teapot'))
    v = fl['v'].reshape(-1, 3)
    assert np.allclose(v.min(0), [-14.9985, -9.999, 0.0], atol=1e-3)
    assert np.allclose(v.max(0), [17.1389, 9.999, 15.7484], atol=1e-3)
    assert int((thmodel_ref.tri_area(fl['v']) == 0).sum()) == 32


def test_cornell_light_quad_and_walls():
    fl = thmodel_ref.load_flat(model_path('cornell-box'))
    light = fl['v'][fl['materials'][fl['mat_idx'], 0] > 0]
    assert light.shape == (2, 3, 3)
    assert np.allclose(light[:, :, 2], 19.921038, atol=1e-5)
    assert np.allclose(np.abs(light[:, :, :2]).max(), 3.495876, atol=1e-5)
    assert abs(thmodel_ref.tri_area(light).sum() - 48.8846) < 1e-3
    n = np.cross(light[:, 1] - light[:, 0], light[:, 2] - light[:, 0])
    assert (n[:, 2] < 0).all() and np.allclose(n[:, :2], 0)          # faces down into the room
    red = fl['v'][[n == 'Red' for n in np.array(fl['names'])[fl['mat_idx']]]]
    green = fl['v'][[n == 'Green' for n in np.array(fl['names'])[fl['mat_idx']]]]
    assert np.allclose(red[:, :, 0], -10, atol=1e-4) and np.allclose(green[:, :, 0], 10, atol=1e-4)


@pytest.mark.parametrize('name', sorted(KAT))
def test_flattened_scene_properties(name):
    fl = thmodel_ref.load_flat(model_path(name))
    assert len(fl['v']) == sum(KAT[name][1])
    area = thmodel_ref.tri_area(fl['v'])
    assert abs(area.sum() - AREAS[name]) < 1e-2
    # geometric normal (v1-v0)x(v2-v0) agrees in sign with the stored vertex normals on every non-degenerate triangle
    g = np.cross(fl['v'][:, 1] - fl['v'][:, 0], fl['v'][:, 2] - fl['v'][:, 0]).astype(np.float64)
    ok = area > 0
    s = np.einsum('tj,tkj->tk', g[ok], fl['n'][ok].astype(np.float64))
    assert (s > 0).all()
    assert np.allclose(np.linalg.norm(fl['n'], axis=2)[ok], 1.0, atol=1e-5)


@pytest.mark.parametrize('name', sorted(KAT))
def test_cpp_reader_matches_restatement(name):
    pt = P.PathTracer(8, 8, device=-1)
    pt.Scene(model_path(name))
    got, ref = pt.get_scene(), thmodel_ref.load_flat(model_path(name))
    assert np.array_equal(got['v'].view(np.uint32), ref['v'].view(np.uint32))        # positions bit-exact
    assert np.abs(got['n'] - ref['n']).max() <= 1.2e-7                               # normals to 1 ulp (fma emulation)
    assert np.array_equal(got['mat_idx'], ref['mat_idx'])
    assert np.array_equal(got['materials'], ref['materials'])
    st = pt.stats()
    assert st.numTriangles == sum(KAT[name][1])
    assert st.numLightGroups == (1 if name == 'cornell-box' else 0)


def test_model_matrix_identity_keeps_collada_rotation():
    """Scene(path, matFunc, mat4.Identity): the root node's rotation is applied (scene.d:257)."""
    pt = P.PathTracer(8, 8, device=-1)
    pt.Scene(model_path('teapot'), modelMatrix=np.eye(4, dtype=np.float32))
    got = pt.get_scene()['v']
    raw = thmodel_ref.load_flat(model_path('teapot'))['v']
    # row-vector convention: (x, y, z) -> (x, z, -y)
    assert np.array_equal(got[..., 0], raw[..., 0]) and np.array_equal(got[..., 1], raw[..., 2])
    assert np.array_equal(got[..., 2], -raw[..., 1])
    ref = thmodel_ref.flatten(thmodel_ref.read_thmodel(model_path('teapot')), np.eye(4, dtype=np.float32).reshape(16))
    assert np.array_equal(got.view(np.uint32), ref['v'].view(np.uint32))


def test_citymap_material_table():
    """fillMaterialCitymap, rendering.d:195-244, through upload of named materials is not reachable without a
    citymap model; the table itself is checked through cornell's names, which it maps to the default grey."""
    pt = P.PathTracer(8, 8, device=-1)
    pt.Scene(model_path('cornell-box'), matFunc=P.FILL_MATERIAL_CITYMAP)
    mats = pt.get_scene()['materials']
    assert np.allclose(mats, np.array([[0, 0.7, 0.7, 0.7]] * 4, np.float32))
    assert pt.stats().numLightGroups == 0


def test_reader_errors(tmp_path):
    pt = P.PathTracer(8, 8, device=-1)
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(tmp_path / 'nope.thModel'))
    assert e.value.code == -4 and 'does not exist' in str(e.value)              # modelloader.d:132-135
    bad = tmp_path / 'bad.thModel'
    bad.write_bytes(b'\x04tree' + b'\x00' * 32)
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(bad))
    assert e.value.code == -4 and 'not a thModel format' in str(e.value)        # modelloader.d:139-142
    data = open(model_path('chest1'), 'rb').read()
    trunc = tmp_path / 'trunc.thModel'
    trunc.write_bytes(data[:700])
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(trunc))
    assert e.value.code == -4
    v3 = bytearray(data); v3[12] = 3                                             # version field
    newer = tmp_path / 'v3.thModel'; newer.write_bytes(bytes(v3))
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(newer))
    assert 'old format' in str(e.value)                                          # modelloader.d:144-147
    with pytest.raises(ValueError):
        thmodel_ref.read_thmodel(str(trunc))


def test_malformed_node_graphs_and_counts_are_refused(tmp_path):
    """A file whose node graph is not a tree (the reference's findLeafs, scene.d:213-225, and its parent walk,
    scene.d:258-262, would not terminate on it), or whose element counts exceed what the chunk holds, gives PT_E_IO
    instead of a hang, a stack overflow or a huge allocation."""
    import struct
    pt = P.PathTracer(8, 8, device=-1)
    data = open(model_path('chest1'), 'rb').read()
    # node 0: parent none, no meshes, one child (node 1)
    pat = struct.pack('<IIII', 0xFFFFFFFF, 0, 1, 1)
    at = data.index(pat)
    loop = bytearray(data); loop[at + 12:at + 16] = struct.pack('<I', 0)           # node 0 lists itself as its child
    p = tmp_path / 'childloop.thModel'; p.write_bytes(bytes(loop))
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(p))
    assert e.value.code == -4 and 'not a tree' in str(e.value)
    # node 1 (the file's last 16 bytes: parent, mesh count, mesh index, child count) names itself as its parent
    assert data[-16:] == struct.pack('<IIII', 0, 1, 0, 0)
    par = bytearray(data); par[-16:-12] = struct.pack('<I', 1)
    p = tmp_path / 'parentloop.thModel'; p.write_bytes(bytes(par))
    with pytest.raises(P.PtError) as e:
        pt.Scene(str(p))
    assert e.value.code == -4 and 'cycle' in str(e.value)
    # counts far beyond the payload: textures, vertices, faces, nodes
    for marker, skip in ((b'\x08textures', 4), (b'\x08vertices', -4), (b'\x05faces', 4), (b'\x05nodes', 4)):
        at = data.index(marker) + len(marker) + skip          # the count follows the chunk length (vertices: precedes the chunk)
        if skip < 0:
            at = data.index(marker) - 4
        big = bytearray(data); big[at:at + 4] = struct.pack('<I', 0xFFFFFFF0)
        p = tmp_path / 'big.thModel'; p.write_bytes(bytes(big))
        with pytest.raises(P.PtError) as e:
            pt.Scene(str(p))
        assert e.value.code == -4, marker
