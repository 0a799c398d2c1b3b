"""Multi-GPU layer of the FFT hot path: one process per GPU, torch.distributed over RCCL (backend
"nccl" on ROCm) or gloo (CPU tests).  SURVEY 8(e):

  * batched cfft / rfft / q15 / q31 and large-N transforms are independent -> contiguous batch
    slices per rank, NO data-path collective (``batch_slice``);
  * GNSS acquisition: scenes are sharded the same way; inside ONE scene the Doppler bins are
    sharded (``doppler_range``: the unit ``gnss_acquire_dev`` takes) and the per-PRN winners of
    every rank are merged with ONE small collective (``merge_best``): an all-gather of the
    n_prn x 32-byte tuples (1 KiB per rank for 32 PRNs), then the same rule the single-GPU
    kernel applies -- largest peak, lowest Doppler bin on ties -- evaluated identically on every
    rank.  Latency-bound over xGMI; nothing else crosses the links.

Nothing here computes transforms: the kernels live in libarmfft_mi355.so.
"""
import numpy as np

ACQ_WORDS = 8  # a gnss_acq_result is eight 32-bit words


def batch_slice(total, rank, world):
    """Contiguous [begin, end) share of `total` independent transforms (or scenes) for `rank`;
    the first total % world ranks take one extra."""
    if world < 1 or not 0 <= rank < world or total < 0:
        raise ValueError("bad sharding request: total=%r rank=%r world=%r" % (total, rank, world))
    q, r = divmod(total, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def doppler_range(n_dop, rank, world):
    """[dop_begin, dop_end) of the plan's Doppler list searched by `rank` (may be empty)."""
    return batch_slice(n_dop, rank, world)


def merge_best_rows(rows):
    """rows: int32 array [world, n_prn, 8] of gnss_acq_result tuples (peak < 0 marks "rank had no
    bins").  Returns [n_prn, 8]: per PRN the tuple with the largest peak, lowest doppler_bin on ties."""
    rows = np.asarray(rows)
    peak = rows[..., 0].view(np.float32)
    dbin = rows[..., 4]
    # order-preserving key: peaks are magnitudes (>= 0) or the -1 marker
    key = np.where(peak >= 0, peak.view(np.int32).astype(np.int64) + 1, 0) << 32
    key = key | (0x7FFFFFFF - dbin.astype(np.int64))
    win = np.argmax(key, axis=0)
    return np.take_along_axis(rows, win[None, :, None], axis=0)[0]


def merge_best(best, group=None):
    """best: torch int32 tensor [n_prn, 8] (this rank's winners over ITS Doppler bins, on the
    device the process group works on; set peak = -1 where the rank searched no bin).
    Returns the merged [n_prn, 8] tensor, identical on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return best
    world = dist.get_world_size(group)
    src = best.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        src = src.cpu()                                # gloo gathers host tensors (rehearsals of the multi-GPU path without RCCL)
    gathered = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(gathered, src, group=group)
    rows = torch.stack(gathered).to(best.device)       # [world, n_prn, 8]
    peak = rows[..., 0].view(torch.float32)
    key = torch.where(peak >= 0, rows[..., 0].to(torch.int64) + 1, torch.zeros_like(rows[..., 0], dtype=torch.int64)) << 32
    key = key | (0x7FFFFFFF - rows[..., 4].to(torch.int64))
    win = key.argmax(dim=0)                            # [n_prn]
    return torch.gather(rows, 0, win[None, :, None].expand(1, rows.shape[1], ACQ_WORDS))[0]


def sharded_acquire(acquire, n_dop, n_rows, rank=None, world=None, row_prns=None, device=None, group=None):
    """One Doppler-sharded acquisition step, as `bench.py --gnss-shard doppler` and a multi-GPU
    caller of gnss_acquire_dev run it: this rank searches its share [d0, d1) of the plan's n_dop
    bins through `acquire(d0, d1)` -- which returns the rank's winners as an int32 tensor
    [n_rows, 8] (n_rows = scenes x PRNs; doppler_bin holds GLOBAL bin indices, as
    gnss_acquire_dev writes them) -- and the winners of all ranks are merged by ONE all-gather
    (`merge_best`).  A rank whose share is empty contributes `empty_best` rows and still takes
    part in the collective.  Returns (merged [n_rows, 8], (d0, d1)); merged is identical on
    every rank and equal to the single-rank search over all bins."""
    import torch.distributed as dist

    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
    d0, d1 = doppler_range(n_dop, rank, world)
    mine = acquire(d0, d1) if d1 > d0 else empty_best(n_rows, row_prns, device)
    if tuple(mine.shape) != (n_rows, ACQ_WORDS):
        raise ValueError("acquire() must return an int32 tensor [%d, %d], got %r" % (n_rows, ACQ_WORDS, tuple(mine.shape)))
    return merge_best(mine, group), (d0, d1)


def empty_best(n_prn, prns=None, device=None):
    """The tuple list of a rank that searched nothing (peak = -1, indices -1)."""
    import torch
    t = torch.zeros((n_prn, ACQ_WORDS), dtype=torch.int32, device=device)
    t[:, 0] = torch.tensor(-1.0).view(torch.int32)
    t[:, 1] = torch.tensor(-1e10).view(torch.int32)
    t[:, 2:5] = -1
    if prns is not None:
        t[:, 5] = torch.as_tensor(prns, dtype=torch.int32, device=device)
    return t
