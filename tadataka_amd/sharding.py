"""Pair sharding across GPUs: independent frame pairs are dealt to
ranks in contiguous blocks, every rank estimates its own block with no data-path
collective, and the recovered poses are all-gathered at the end (RCCL over xGMI
when the process group is `nccl`; `gloo` in the CPU tests).

torch is imported lazily: the single-GPU path never needs it."""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [begin, end) of `n_items` for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pair_seeds(rank, pairs_per_rank):
    """Global pair ids (= synthetic seeds) owned by `rank` under weak scaling."""
    return np.arange(rank * pairs_per_rank, (rank + 1) * pairs_per_rank)


def _tensor(array, device):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64))
    return t.to(device) if device is not None else t


def all_gather_poses(local_poses, dist=None, device=None):
    """[B, 12] poses of this rank -> [world * B, 12] in rank order on every rank.
    Every rank must contribute the same B (weak scaling)."""
    local_poses = np.ascontiguousarray(local_poses, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return local_poses.copy()
    import torch
    mine = _tensor(local_poses, device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return torch.cat(parts, dim=0).cpu().numpy()


class PoseGather(object):
    """All-gather of the per-rank [B, 12] poses that does not stall the rank:
    start() queues the collective (async_op) and returns at once, finish() waits
    for the oldest one.  bench.py starts the gather of step k and collects it
    after the estimation of step k + 1, so the RCCL launch and the two small
    copies hide under the next batch's kernels.  Buffers are allocated once."""

    def __init__(self, pairs_per_rank, dist=None, device=None):
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.device = device
        self.pending = []
        if self.dist is not None:
            import torch
            self.world = self.dist.get_world_size()
            on_gpu = device is not None
            self.mine = torch.empty((pairs_per_rank, 12), dtype=torch.float64, device=device)
            self.out = torch.empty((self.world * pairs_per_rank, 12), dtype=torch.float64, device=device)
            # pinned staging on the GPU path: both copies are asynchronous, one event wait in finish()
            self.mine_host = torch.empty((pairs_per_rank, 12), dtype=torch.float64, pin_memory=on_gpu)
            self.out_host = torch.empty_like(self.out, device="cpu", pin_memory=on_gpu)
            self.done = torch.cuda.Event() if on_gpu else None

    def start(self, local_poses):
        local_poses = np.ascontiguousarray(local_poses, dtype=np.float64)
        if self.dist is None:
            self.pending.append(local_poses.copy())
            return
        import torch
        if self.pending:               # one set of buffers: at most one gather in flight
            raise RuntimeError("finish() the previous gather first")
        self.mine_host.copy_(torch.from_numpy(local_poses))
        self.mine.copy_(self.mine_host, non_blocking=True)
        work = self.dist.all_gather_into_tensor(self.out, self.mine, async_op=True)
        if self.done is not None:
            work.wait()                # orders the current torch stream after the collective; does not block the host
            self.out_host.copy_(self.out, non_blocking=True)
            self.done.record()
        self.pending.append(work)

    def finish(self):
        """Poses of all ranks, [world * B, 12] in rank order, of the oldest start()."""
        item = self.pending.pop(0)
        if self.dist is None:
            return item
        if self.done is not None:
            self.done.synchronize()
            return self.out_host.numpy().copy()
        item.wait()
        return self.out.numpy().copy()


def reduce_scalars(values, op, dist=None, device=None):
    """Element-wise MAX or SUM of a few float64 scalars over all ranks."""
    values = np.asarray(values, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return values.copy()
    t = _tensor(values, device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.cpu().numpy()
