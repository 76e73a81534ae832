"""Pair sharding across GPUs: independent frame pairs are dealt
to ranks in contiguous blocks, every rank estimates its own block with no
data-path collective, and the recovered poses are all-gathered at the end --
ncclAllGather from RCCL over xGMI through the C ABI (tdk_comm_*,
include/tadataka_hip.h), one process per GPU.  No PyTorch anywhere.

A communicator is anything with `rank`, `world`, `all_gather(array)`,
`all_reduce(values, op)` and `barrier()`: LocalComm (one process), RcclComm (the
product path) or the gloo adapter the CPU tests bring along."""
import ctypes as C
import os
import time

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


class LocalComm(object):
    """World of one: every collective is the identity."""
    rank, world = 0, 1
    kind = "local"

    def all_gather(self, array):
        return np.array(array, dtype=np.float64)

    def all_reduce(self, values, op):
        return np.array(values, dtype=np.float64)

    def barrier(self):
        pass

    def close(self):
        pass


class _StdoutToStderr(object):
    """RCCL prints a version banner on the C-level stdout when a communicator comes up; bench.py
    owes its caller exactly one JSON line there.  While the communicator is created, file
    descriptor 1 points at stderr (C stdio flushed on both sides of the switch)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._libc = C.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class RcclComm(object):
    """RCCL communicator of this process (tdk_comm).  The device must have been
    selected (tdk_set_device) before; creation is collective."""
    kind = "rccl"

    def __init__(self, rank, world, unique_id):
        from tadataka_amd import _lib
        self._lib = _lib
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        with _StdoutToStderr():
            _lib.call("tdk_comm_create", buf, self.rank, self.world, C.byref(self._h))

    @staticmethod
    def unique_id():
        from tadataka_amd import _lib
        buf = (C.c_uint8 * 128)()
        with _StdoutToStderr():
            _lib.call("tdk_comm_unique_id", buf)
        return bytes(buf)

    def all_gather(self, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        out = np.empty((self.world,) + a.shape)
        self._lib.call("tdk_comm_all_gather", self._h, a.ctypes.data_as(self._lib.c_double_p), a.size,
                       out.ctypes.data_as(self._lib.c_double_p))
        return out.reshape((self.world * a.shape[0],) + a.shape[1:]) if a.ndim else out

    def all_reduce(self, values, op):
        v = np.array(values, dtype=np.float64)
        self._lib.call("tdk_comm_all_reduce", self._h, v.ctypes.data_as(self._lib.c_double_p), v.size,
                       {"sum": 0, "max": 1}[op])
        return v

    def barrier(self):
        self._lib.call("tdk_comm_barrier", self._h)

    # device-resident pose gather, asynchronous on the batch's stream
    def gather_poses_start(self, batch):
        self._lib.call("tdk_dvo_gather_poses_start", batch._h, self._h)
        self._pending_shape = (self.world * batch.n_pairs, 12)

    def gather_poses_finish(self):
        out = np.empty(self._pending_shape)
        self._lib.call("tdk_dvo_gather_poses_finish", self._h, out.ctypes.data_as(self._lib.c_double_p))
        return out

    def close(self):
        if self._h:
            self._lib.call("tdk_comm_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FileComm(object):
    """Last-resort exchange through files in TMPDIR for the FEW BYTES this path ever moves
    between ranks (poses, a handful of scalars): used only when RCCL cannot be brought up, so that
    a multi-GPU measurement still completes -- and says so (`kind`).  The estimation itself never
    goes through here; there is no data-path collective to fall back from."""
    kind = "file"

    def __init__(self, rank, world, key):
        self.rank, self.world = int(rank), int(world)
        self._dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "tdk_filecomm_%s" % key)
        os.makedirs(self._dir, exist_ok=True)
        self._seq = 0

    def _exchange(self, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        self._seq += 1
        mine = os.path.join(self._dir, "%d_%d.npy" % (self._seq, self.rank))
        np.save(mine + ".tmp.npy", a)
        os.replace(mine + ".tmp.npy", mine)
        parts, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self._dir, "%d_%d.npy" % (self._seq, r))
            while not os.path.exists(path):
                if time.time() - t0 > 600.0:
                    raise RuntimeError("file exchange: rank %d never arrived" % r)
                time.sleep(0.0005)
            parts.append(np.load(path))
        if self._seq > 2:                           # everybody has passed exchange seq - 2 by now
            try:
                os.unlink(os.path.join(self._dir, "%d_%d.npy" % (self._seq - 2, self.rank)))
            except OSError:
                pass
        return parts

    def all_gather(self, array):
        parts = self._exchange(array)
        return np.concatenate([p.reshape((-1,) + p.shape[1:]) if p.ndim else p.reshape(1) for p in parts], axis=0)

    def all_reduce(self, values, op):
        parts = np.array(self._exchange(np.asarray(values, dtype=np.float64).reshape(-1)))
        return parts.max(axis=0) if op == "max" else parts.sum(axis=0)

    def barrier(self):
        self._exchange(np.zeros(1))

    def close(self):
        """Removes this rank's files (the peers have read them once they passed the same barrier)."""
        try:
            self.barrier()
        except RuntimeError:
            pass
        for seq in range(max(1, self._seq - 2), self._seq):          # everything but the closing barrier's file
            try:
                os.unlink(os.path.join(self._dir, "%d_%d.npy" % (seq, self.rank)))
            except OSError:
                pass


def _rendezvous_path():
    # all ranks of one launch share the launcher as parent and the rendezvous port
    key = os.environ.get("TDK_RENDEZVOUS_KEY") or "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "tdk_rccl_%s.id" % key)


def connect_or_fallback(rank=None, world=None):
    """connect(); if RCCL cannot be initialised, a FileComm and the reason -- (comm, error or None)."""
    try:
        return connect(rank, world), None
    except Exception as e:                          # noqa: BLE001  (library missing, bootstrap failure, ...)
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        key = "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        return FileComm(rank, world, key), repr(e)


def connect(rank=None, world=None, timeout=300.0):
    """The communicator of this process from the launcher's environment (RANK,
    WORLD_SIZE -- what torch.distributed.run and bench.py's own spawner export).
    Rank 0 publishes the RCCL unique id in a file keyed by the rendezvous port and
    the launcher's pid; the others wait for it.  One process: LocalComm."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if world <= 1:
        return LocalComm()
    path = _rendezvous_path()
    if rank == 0:
        uid = RcclComm.unique_id()
        tmp = path + ".%d.tmp" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)           # atomic: readers see nothing or all 128 bytes
    else:
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d: no RCCL unique id at %s after %.0f s" % (rank, path, timeout))
            time.sleep(0.02)
    comm = RcclComm(rank, world, uid)   # collective (ncclCommInitRank): returns once every rank has joined
    comm.barrier()                      # ... and every rank has therefore read the id
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    return comm


def all_gather_poses(local_poses, comm=None):
    """[B, 12] poses of this rank -> [world * B, 12] in rank order on every rank.
    Every rank must contribute the same B (weak scaling)."""
    local_poses = np.ascontiguousarray(local_poses, dtype=np.float64)
    if comm is None or comm.world == 1:
        return local_poses.copy()
    return comm.all_gather(local_poses)


class PoseGather(object):
    """All-gather of the per-rank [B, 12] poses that does not stall the rank:
    start() queues the collective and returns at once, finish() waits for the
    oldest one.  bench.py starts the gather of step k and collects it after the
    estimation of step k + 1, so the RCCL launch and the copy hide under the next
    batch's kernels.  With an RcclComm and a DvoBatch the gather reads the poses
    where the device loop left them (ncclAllGather on the batch's stream)."""

    def __init__(self, pairs_per_rank, comm=None):
        self.comm = comm if (comm is not None and comm.world > 1) else None
        self.pairs_per_rank = pairs_per_rank
        self.pending = []

    def start(self, local_poses, batch=None):
        local_poses = np.ascontiguousarray(local_poses, dtype=np.float64)
        if self.comm is None:
            self.pending.append(local_poses.copy())
            return
        if self.pending:                       # one set of buffers: at most one gather in flight
            raise RuntimeError("finish() the previous gather first")
        if batch is not None and hasattr(self.comm, "gather_poses_start"):
            self.comm.gather_poses_start(batch)
            self.pending.append(None)          # collected in finish()
        else:
            self.pending.append(self.comm.all_gather(local_poses))

    def finish(self):
        """Poses of all ranks, [world * B, 12] in rank order, of the oldest start()."""
        item = self.pending.pop(0)
        if item is None:
            return self.comm.gather_poses_finish()
        return item


def reduce_scalars(values, op, comm=None):
    """Element-wise MAX or SUM of a few float64 scalars over all ranks."""
    values = np.asarray(values, dtype=np.float64)
    if comm is None or comm.world == 1:
        return values.copy()
    return comm.all_reduce(values, op)
