"""The N > 1 path on CPU: two processes, gloo backend, the same share-out and combine code that bench.py runs
over RCCL.  The per-rank renderer here is a deterministic stand-in that fills (sum.rgb, n) as a function of
(pixel, sample) -- this test is about partition + collective, not about radiance."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import multigpu

W, H, SPP, BAND_ROWS = 40, 23, 6, 4


def fake_sample(pixel, sample):
    """Stand-in radiance of one (pixel, sample): exact in fp32 so sums compare bit for bit."""
    return np.stack([(pixel % 7 + sample).astype(np.float32), (pixel % 5).astype(np.float32) * 0.5,
                     ((pixel + sample) % 3).astype(np.float32)], axis=1)


class FakeTracer:
    """Implements the three calls multigpu.render_share uses, writing into a CPU accumulator."""

    def __init__(self, acc):
        self.acc = acc.view(-1, 4)
        self.sample_base, self.sample_stride = 0, 1
        self.drawn = []                       # sample indices drawn for pixel 0, in order

    def set_config(self, sample_base=0, sample_stride=1):
        self.sample_base, self.sample_stride = sample_base, sample_stride

    def _add(self, pixels, spp):
        a = self.acc.numpy()
        for s in range(spp):
            # the library numbers a pixel's sample by what its accumulator already holds (PtConfig.sampleBase / sampleStride)
            idx = self.sample_base + a[pixels, 3].astype(np.int64) * self.sample_stride
            if len(pixels) and pixels[0] == 0:
                self.drawn.append(int(idx[0]))
            a[pixels, :3] += fake_sample(pixels, idx)
            a[pixels, 3] += 1

    def render_frame(self, spp):
        self._add(np.arange(W * H), spp)

    def render_bands(self, band_rows, first, stride, spp):
        rows = multigpu.rows_of_rank(H, band_rows, first, stride)
        px = (np.array(rows)[:, None] * W + np.arange(W)[None, :]).reshape(-1)
        self._add(px, spp)


def expected(split, world):
    acc = torch.zeros(W * H * 4)
    t = FakeTracer(acc)
    if split == 'bands':
        t.render_frame(SPP)
    else:
        t.render_frame(SPP * world)           # indices 0 .. SPP * world - 1, each once
    return acc


def worker(rank, world, port, split, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert multigpu.rank_world() == (rank, world)
        acc = torch.zeros(W * H * 4)
        multigpu.render_share(FakeTracer(acc), SPP, rank, world, split=split, band_rows=BAND_ROWS)
        own = acc.clone()
        if split == 'bands':                        # disjoint rows: gathered; the sample split is reduced
            multigpu.gather_bands(acc, W, H, BAND_ROWS, rank, world, dist, dst=0)
        else:
            multigpu.combine(acc, dist, dst=0)
        if rank == 0:
            q.put((split, acc.numpy().copy(), own.numpy().copy()))
        both = own.clone()
        multigpu.combine(both, dist, all_ranks=True)
        assert torch.equal(both, torch.from_numpy(expected(split, world).numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize('split', ['bands', 'samples'])
def test_two_ranks_gloo(split):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, split, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, combined, own0 = got
    exp = expected(split, world).numpy()
    assert np.array_equal(combined, exp)                                       # bit-identical to one process
    n = combined.reshape(-1, 4)[:, 3]
    assert (n == (SPP if split == 'bands' else SPP * world)).all()
    if split == 'bands':                                                       # rank 0 alone holds only its bands
        rows0 = set(multigpu.rows_of_rank(H, BAND_ROWS, 0, world))
        n0 = own0.reshape(H, W, 4)[:, :, 3]
        for r in range(H):
            assert (n0[r] == (SPP if r in rows0 else 0)).all()


def test_single_process_combine_is_identity():
    acc = torch.arange(8, dtype=torch.float32)
    assert multigpu.combine(acc, None) is acc
    with pytest.raises(ValueError):
        multigpu.render_share(FakeTracer(torch.zeros(W * H * 4)), 1, 0, 1, split='tiles')


def test_sample_split_draws_disjoint_indices_over_several_passes():
    """Progressive rendering with the sample split: each rank keeps its own accumulator, and over P passes the G ranks
    together draw every index in [0, P * G * spp) exactly once (a slice [rank * spp, (rank + 1) * spp) per pass would
    replay indices from the second pass on)."""
    world, passes = 3, 4
    drawn = []
    for rank in range(world):
        t = FakeTracer(torch.zeros(W * H * 4))
        for _ in range(passes):
            multigpu.render_share(t, SPP, rank, world, split='samples')
        assert len(t.drawn) == passes * SPP and all(i % world == rank for i in t.drawn)
        drawn += t.drawn
    assert sorted(drawn) == list(range(passes * world * SPP))


def test_native_band_bookkeeping_and_host_only_group():
    """ptgpu_band_rows (the rows GPU i of G renders and sends) against the Python share-out, and a group without
    devices: scene and camera reach every rank, rendering fails loudly."""
    import pathtracer_amd as P
    from conftest import model_path
    for height, rows, world in ((23, 4, 2), (1080, 4, 8), (240, 4, 3), (7, 16, 4), (64, 1, 5)):
        seen = []
        for rank in range(world):
            got = P.band_rows(height, rows, rank, world)
            assert list(got) == multigpu.rows_of_rank(height, rows, rank, world)
            seen += list(got)
        assert sorted(seen) == list(range(height))                      # every scanline exactly once
    with pytest.raises(P.PtError):
        P.band_rows(10, 4, 2, 2)
    g = P.GpuGroup(32, 24, [-1, -1, -1], root=1)
    assert g.info() == {'devices': 3, 'root': 1, 'rccl': False}
    g.Scene(model_path('cornell-box'))
    g.set_camera(P.Camera(30.0, 24 / 32).setTransform((-1, 26.5, 10), (0, 0, 9), (0, 0, 1)))
    for r in range(3):
        st = g.rank(r).stats()
        assert st.numTriangles == 52 and st.numLightGroups == 1
    with pytest.raises(P.PtError) as e:
        g.render(1)
    assert e.value.code == -3 and 'GPU 0' in str(e.value)
    with pytest.raises(P.PtError):
        g.combine()
    with pytest.raises(P.PtError):
        P.GpuGroup(32, 24, [-1, 0])
    g.close()
