"""Data-parallel inference over the GPUs of one node: one process per GPU, frames sharded across
ranks, weights replicated, no data-path collective except ONE all-gather per step of the final
detection rows - (B_rank, K, 54) fp32 post-processed boxes = 346 KB per rank at bs=16 (or the (B_rank, K,
33) decoded rows) - over RCCL/xGMI (backend "nccl" is RCCL on ROCm).

Replaces the reference's evaluation-time exchange, which all-gathers every raw head map plus the
whole input batch (model/progressBar.py:85-91, 177-183: ~10.7 MB per frame); frames are independent
(eval-mode BN, per-image frustum association and decode), so sharding is exact: the gathered result
equals a single-GPU run on the concatenated batch bit for bit (tests/test_gpu_model.py::
test_batch_sharding_is_bit_exact, tests/test_distributed_cpu.py).
"""
import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous, balanced split of n_frames over `world` ranks -> (lo, hi) of `rank`."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_detections(det: torch.Tensor, group=None, force_collective=False, async_op=False):
    """(B_rank, K, W) per rank -> (sum B_rank, K, W) on every rank, rank-major.

    Equal shards take the single-buffer all_gather_into_tensor path (one RCCL launch, latency
    bound: ~1.4 us of wire time per peer link at 211 KB); ragged shards are padded to the largest.
    force_collective: issue the collective also at world size 1 (tests: the RCCL call path itself runs
    on a one-GPU box).  async_op (equal shards only): return (out, work) without waiting - the gather
    runs on RCCL's own stream beside the caller's next launches; `work.wait()` orders the current stream
    behind it (DetectionGatherer wraps this)."""
    if not dist.is_available() or not dist.is_initialized():
        if async_op:
            return det, None
        return det
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return (det, None) if async_op else det
    det = det.contiguous()
    if getattr(gather_detections, "_assume_equal", False) or world == 1:
        counts = [det.shape[0]] * world
    else:
        sizes = torch.tensor([det.shape[0]], device=det.device, dtype=torch.int64)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    bmax = max(counts)
    if all(c == bmax for c in counts):
        out = torch.empty((world * bmax,) + tuple(det.shape[1:]), device=det.device, dtype=det.dtype)
        work = dist.all_gather_into_tensor(out, det, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError("async gather needs equal shards (assume_equal_shards)")
    pad = torch.zeros((bmax,) + tuple(det.shape[1:]), device=det.device, dtype=det.dtype)
    pad[:det.shape[0]] = det
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


class _Pending:
    def __init__(self, out, work, keep):
        self.out, self.work, self.keep = out, work, keep

    def wait(self):
        """Order the current stream behind the gather (no host block on RCCL) and return the gathered rows."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.keep = None
        return self.out


class DetectionGatherer:
    """The per-step all-gather overlapped with the next step ("overlap with the next batch's
    backbone"): `submit(det)` enqueues the RCCL all-gather of this step's rows asynchronously - it runs on
    the communicator's stream - and returns at once; the caller launches the next forward and calls
    `.wait()` on the handle when it consumes the gathered boxes.  Equal shards per rank (weak scaling)."""

    def __init__(self, device=None, group=None, force_collective=False):
        self.group, self.force = group, force_collective

    def submit(self, det: torch.Tensor) -> _Pending:
        out, work = gather_detections(det, self.group, self.force, async_op=True)
        return _Pending(out, work, det)


def assume_equal_shards(flag=True):
    """Skip the per-step shard-size exchange when every rank is known to hold the same batch
    (the benchmark / weak-scaling case): one collective per step instead of two."""
    gather_detections._assume_equal = bool(flag)
