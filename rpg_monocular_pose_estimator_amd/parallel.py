"""Multi-GPU sharding of the hot path: frames are independent in brute-force-every-frame mode
(pose_estimator.cpp:68-91 reads no state when it_since_initialized_ < 1), so a batch is split into
contiguous chunks, one per rank (one process per GPU), and the only collective is the gather of
the fixed-size per-frame pose records (432 B/frame) — RCCL over xGMI on GPUs (backend "nccl"),
gloo on CPU for the tests."""
import numpy as np

from .binding import RESULT_DTYPE


def shard_bounds(n_frames, rank, world):
    """Contiguous chunk [lo, hi) of rank `rank`: sizes differ by at most one frame (the same rule as the C
    ABI's mpe_shard_bounds, which mpe_estimate_batch_multi uses for its per-device shards)."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local_bytes, world, out=None):
    """all_gather of equally sized uint8 record buffers (torch tensors, CPU/gloo or CUDA/nccl): every rank
    ends up with every record.  Costs world x the traffic of gather_records_to_root."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_bytes
    if out is None:
        out = torch.empty(world * local_bytes.numel(), dtype=torch.uint8, device=local_bytes.device)
    dist.all_gather_into_tensor(out, local_bytes)
    return out


def gather_records_to_root(local_bytes, rank, world, out=None, dst=0, async_op=False, force=False):
    """The per-rank pose gather: rank `dst` receives the record buffers of all ranks (in rank order) into `out`
    (world * len bytes; allocated if None), the others only send theirs — point-to-point traffic over xGMI,
    432 B per frame and rank.  async_op=True returns (out, work): the transfer then runs beside the next batch's
    kernels; call work.wait() before reading `out` on `dst` / before overwriting `local_bytes`."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:  # (force: the collective itself on a one-rank group — exercises the RCCL path on one GPU)
        return (local_bytes, None) if async_op else local_bytes
    pieces = None
    if rank == dst:
        if out is None:
            out = torch.empty(world * local_bytes.numel(), dtype=torch.uint8, device=local_bytes.device)
        pieces = list(out.view(world, -1).unbind(0))
    work = dist.gather(local_bytes, pieces, dst=dst, async_op=async_op)
    return (out, work) if async_op else out


class RootGatherPipeline:
    """Double-buffered, asynchronous gather of per-step record buffers to rank `dst` (what bench.py does every
    step): step k writes its records into `local(k)`, `submit(k)` starts the transfer, and the kernels of step
    k+1 — which write the OTHER local buffer — run beside it.  A buffer is only handed out again after its
    previous transfer has been waited for.  On `dst`, `gathered(k)` is valid after `wait(k)` / `finish()`."""

    def __init__(self, rank, world, nbytes, device, dst=0, force_collective=False):
        import torch
        self.rank, self.world, self.dst = rank, world, dst
        self.collective = world > 1 or force_collective  # (forced: a one-rank group still runs the gather)
        n = 2  # (also at world == 1: two submissions may be in flight, each with its own record buffer)
        self._local = [torch.zeros(nbytes, dtype=torch.uint8, device=device) for _ in range(n)]
        self._out = [torch.zeros(world * nbytes, dtype=torch.uint8, device=device) if (self.collective and rank == dst)
                     else None for _ in range(n)]
        self._work = [None] * n

    def _slot(self, step):
        return step % len(self._local)

    def local(self, step):
        """Result buffer for `step`; waits (stream-level on CUDA) until its previous transfer has left."""
        self.wait(step)
        return self._local[self._slot(step)]

    def submit(self, step):
        if not self.collective:
            return
        b = self._slot(step)
        _, self._work[b] = gather_records_to_root(self._local[b], self.rank, self.world, out=self._out[b],
                                                  dst=self.dst, async_op=True, force=True)

    def wait(self, step):
        b = self._slot(step)
        if self._work[b] is not None:
            self._work[b].wait()
            self._work[b] = None

    def finish(self):
        for b in range(len(self._work)):
            self.wait(b)

    def gathered(self, step):
        return self._out[self._slot(step)] if self.collective else self._local[self._slot(step)]


def records_from_bytes(t):
    """uint8 torch tensor (any device) -> numpy structured array of mpe_result records."""
    return np.frombuffer(t.detach().cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
