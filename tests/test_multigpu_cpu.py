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
        self.sample_base = 0

    def set_config(self, sample_base=0):
        self.sample_base = sample_base

    def _add(self, pixels, spp):
        a = self.acc.numpy()
        for s in range(spp):
            a[pixels, :3] += fake_sample(pixels, np.full(len(pixels), self.sample_base + s))
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
        for r in range(world):
            t.set_config(sample_base=r * SPP); t.render_frame(SPP)
    return acc


def worker(rank, world, port, split, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert multigpu.rank_world() == (rank, world)
        acc = torch.zeros(W * H * 4)
        multigpu.render_share(FakeTracer(acc), SPP, rank, world, split=split, band_rows=BAND_ROWS)
        own = acc.clone()
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
