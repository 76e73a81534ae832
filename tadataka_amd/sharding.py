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


def reduce_scalars(values, op, dist=None, device=None):
    """Element-wise MAX or SUM of a few float64 scalars over all ranks."""
    values = np.asarray(values, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return values.copy()
    t = _tensor(values, device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.cpu().numpy()
