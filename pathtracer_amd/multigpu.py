"""One process per GPU: how a frame is shared out and put back together.

The reference's only parallelism is disjoint 4-scanline tasks over one Pixel[] with a barrier per pass
(main.d:184-192, 273-288).  Across GPUs the same decomposition is used with the scene replicated:

  split='bands'    rank r renders bands r, r+G, r+2G, ... of `band_rows` scanlines, all spp samples.
                   Streams are keyed (seed, pixel, sample), so the frame is bit-identical for any G.
  split='samples'  every rank renders all pixels with its own residue class of sample indices
                   (sampleBase = rank, sampleStride = G: sample k of a rank's pixel draws stream index
                   rank + k * G): G x spp samples per pixel in total (weak scaling).  The classes stay
                   disjoint over any number of passes and across checkpoint / resume, because k counts
                   the samples the RANK's accumulator already holds -- so each rank keeps its own
                   accumulator between passes and combines into a separate frame.

Either way each rank ends with a full-frame (sum.rgb, n) accumulator that is zero where it did no work,
and ONE reduce(sum) over RCCL/xGMI puts the frame together (x + 0 is exact, so the band split stays
bit-identical).  No other communication exists on this path.
"""
import os


def rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def band_share(rank, world):
    """(band_first, band_stride) of a rank: interleaved bands balance scene-dependent cost."""
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    return rank, world


def rows_of_rank(height, band_rows, rank, world):
    """Scanlines a rank owns under the band split (host-side helper, also used by tests)."""
    first, stride = band_share(rank, world)
    nb = (height + band_rows - 1) // band_rows
    rows = []
    for b in range(first, nb, stride):
        rows.extend(range(b * band_rows, min(height, (b + 1) * band_rows)))
    return rows


def render_share(tracer, spp, rank, world, split='bands', band_rows=4):
    """Enqueue this rank's share of the frame on its tracer (asynchronous)."""
    if split == 'bands':
        first, stride = band_share(rank, world)
        tracer.render_bands(band_rows, first, stride, spp)
    elif split == 'samples':
        tracer.set_config(sample_base=rank, sample_stride=world)
        tracer.render_frame(spp)
    else:
        raise ValueError("split must be 'bands' or 'samples'")


def combine(acc, dist=None, dst=0, all_ranks=False):
    """Sum the per-rank accumulators (a torch tensor on the rank's device, or a CPU tensor with gloo).

    dist: the torch.distributed module of an initialised process group, or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return acc
    if all_ranks:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def gather_bands(acc, width, height, band_rows, rank, world, dist, dst=0, via_host=False):
    """Band split: every rank's accumulator holds its own scanlines and zeros elsewhere.  Instead of reducing `world`
    whole frames, each rank sends only its rows (1 / world of the frame) to `dst`, which writes them into its own
    accumulator.  One gather; padded to the largest share so that all pieces have one size.  via_host: move the pieces
    through host memory (gloo rehearsal)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return acc
    frame = acc.view(height, width * 4)
    shares = [rows_of_rank(height, band_rows, r, world) for r in range(world)]
    most = max(len(s) for s in shares)
    dev = 'cpu' if via_host else acc.device
    mine = torch.tensor(shares[rank], dtype=torch.long, device=acc.device)
    piece = torch.zeros(most, width * 4, dtype=acc.dtype, device=dev)
    piece[:len(shares[rank])] = frame.index_select(0, mine).to(dev)
    pieces = [torch.empty_like(piece) for _ in range(world)] if rank == dst else None
    dist.gather(piece, pieces, dst=dst)
    if rank == dst:
        for r in range(world):
            if r != dst and shares[r]:
                idx = torch.tensor(shares[r], dtype=torch.long, device=acc.device)
                frame.index_copy_(0, idx, pieces[r][:len(shares[r])].to(acc.device))
    return acc
