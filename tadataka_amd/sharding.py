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
    def available():
        """Raises unless librccl can be opened and has the symbols the C ABI uses."""
        from tadataka_amd import _lib
        _lib.call("tdk_comm_available")

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


def launch_key():
    """What every rank of ONE launch agrees on without talking: TDK_RENDEZVOUS_KEY if set, else the
    rendezvous port, the launcher's pid (all ranks share the launcher as parent: torch.distributed.run's
    agent or bench.py's own spawner) and the launcher's start time -- so a stale directory of a crashed
    run whose pid and port were reused cannot be mistaken for this one."""
    key = os.environ.get("TDK_RENDEZVOUS_KEY")
    if key:
        return "".join(ch if (ch.isalnum() or ch in "-_.") else "_" for ch in key)
    ppid = os.getppid()
    start = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]      # field 22: starttime
    except (OSError, IndexError):
        pass
    return "%s_%d_%s" % (os.environ.get("MASTER_PORT", "0"), ppid, start)


def rendezvous_dir():
    """Private directory (0700) of this launch under TMPDIR."""
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "tdk_rdv_%s" % launch_key())
    os.makedirs(d, mode=0o700, exist_ok=True)
    # TMPDIR is world-writable and the name is predictable: somebody else's directory (or a symlink to
    # one) under that name must not carry this launch's unique id
    import stat
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid() or (st.st_mode & 0o077):
        raise TransportUnavailable("rendezvous directory %s is not a private directory of this user "
                                   "(owner %d, mode %o)" % (d, st.st_uid, st.st_mode & 0o7777))
    return d


def _publish(directory, name, data):
    """Atomically: readers see nothing or all of `data`.  The file is created exclusively, mode 0600."""
    tmp = os.path.join(directory, ".%s.%d.tmp" % (name, os.getpid()))
    fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
    try:
        os.write(fd, data)
    finally:
        os.close(fd)
    os.replace(tmp, os.path.join(directory, name))


def _collect(directory, names, timeout, what):
    out, t0 = {}, time.time()
    for name in names:
        path = os.path.join(directory, name)
        while True:
            try:
                with open(path, "rb") as f:
                    out[name] = f.read()
                break
            except OSError:
                pass
            if time.time() - t0 > timeout:
                raise RendezvousTimeout("%s: %s did not appear in %s within %.0f s" % (what, name, directory, timeout))
            time.sleep(0.005)
    return out


def _agree(directory, phase, rank, world, error, payload=b"", timeout=300.0):
    """Every rank publishes ok (+ payload) or its error for `phase` and reads everybody's: all ranks
    return the same (errors by rank, payloads by rank) -- the decision what to do next is unanimous."""
    _publish(directory, "%s_%d" % (phase, rank), (b"ok:" + payload) if error is None else b"fail:" + error.encode())
    got = _collect(directory, ["%s_%d" % (phase, r) for r in range(world)], timeout, "rendezvous phase '%s'" % phase)
    errors, payloads = {}, {}
    for r in range(world):
        data = got["%s_%d" % (phase, r)]
        if data.startswith(b"ok:"):
            payloads[r] = data[3:]
        else:
            errors[r] = data[5:].decode("utf-8", "replace")
    return errors, payloads


def _retire(directory, phases, rank, world):
    """After a barrier that follows the last read: remove this rank's files; the last one out removes the directory."""
    for phase in phases:
        try:
            os.unlink(os.path.join(directory, "%s_%d" % (phase, rank)))
        except OSError:
            pass
    try:
        os.rmdir(directory)
    except OSError:
        pass


class TransportUnavailable(RuntimeError):
    """RCCL could not be brought up; every rank of the launch raises this together (the outcome was agreed through
    the rendezvous, see _agree), so a unanimous fallback is possible."""


class RendezvousTimeout(RuntimeError):
    """A peer's file did not appear in time.  NOT a collective outcome: a dead or late peer is seen by each surviving
    rank when ITS OWN time-out fires, so nothing may be decided on it -- it is never turned into a fallback."""


class FileComm(object):
    """Last-resort exchange through files for the FEW BYTES this path ever moves between ranks (poses,
    a handful of scalars): used only when RCCL cannot be brought up AND the ranks share devices (a
    smoke test of the multi-process path on a one-GPU box), so that the run still completes -- and
    says so (`kind`).  The estimation itself never goes through here; there is no data-path collective
    to fall back from."""
    kind = "file"

    def __init__(self, rank, world, directory=None):
        self.rank, self.world = int(rank), int(world)
        self._dir = os.path.join(directory or rendezvous_dir(), "filecomm")
        os.makedirs(self._dir, mode=0o700, exist_ok=True)
        self._seq = 0

    def _exchange(self, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        self._seq += 1
        name = "%d_%d.npy" % (self._seq, self.rank)
        tmp = os.path.join(self._dir, ".%s.tmp.npy" % name)
        np.save(tmp, a)
        os.chmod(tmp, 0o600)
        os.replace(tmp, os.path.join(self._dir, name))
        parts, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self._dir, "%d_%d.npy" % (self._seq, r))
            while not os.path.exists(path):
                if time.time() - t0 > float(os.environ.get("TDK_FILECOMM_TIMEOUT", "600")):
                    raise RuntimeError("file exchange %d: rank %d never arrived" % (self._seq, r))
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
        """A closing barrier, then every rank leaves a `bye` marker -- written only after it has READ the barrier's
        files -- and goes; rank 0 waits for all markers (bounded) and removes what is left.  No rank deletes a file
        a slower peer may still be waiting for."""
        try:
            self.barrier()
        except RuntimeError:
            pass
        try:
            open(os.path.join(self._dir, "bye_%d" % self.rank), "wb").close()
        except OSError:
            pass
        if self.rank != 0:
            return
        t0 = time.time()
        while time.time() - t0 < 30.0:
            if all(os.path.exists(os.path.join(self._dir, "bye_%d" % r)) for r in range(self.world)):
                break
            time.sleep(0.002)
        try:
            for name in os.listdir(self._dir):
                try:
                    os.unlink(os.path.join(self._dir, name))
                except OSError:
                    pass
        except OSError:
            pass
        for d in (self._dir, os.path.dirname(self._dir)):
            try:
                os.rmdir(d)
            except OSError:
                pass


def connect(rank=None, world=None, timeout=300.0):
    """The communicator of this process from the launcher's environment (RANK, WORLD_SIZE -- what
    torch.distributed.run and bench.py's own spawner export).  One process: LocalComm.

    More: the ranks first AGREE, through a private directory of the launch, that every one of them
    can open RCCL (rank 0 adds the unique id to its message); only then do they call
    ncclCommInitRank together, and they agree once more on how that went.  A rank that cannot do its
    part says so instead of leaving the others in a collective that never completes; every rank then
    raises TransportUnavailable with the same list of reasons."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if world <= 1:
        return LocalComm()
    d = rendezvous_dir()
    error, uid = None, b""
    try:
        RcclComm.available()
        if rank == 0:
            uid = RcclComm.unique_id()
    except Exception as e:                          # noqa: BLE001
        error = repr(e)
    errors, payloads = _agree(d, "ready", rank, world, error, uid, timeout)
    if errors:
        _retire_after(d, ("ready",), rank, world, timeout)
        raise TransportUnavailable("RCCL is not available on rank(s) %s" % errors)
    comm, error = None, None
    try:
        comm = RcclComm(rank, world, payloads[0])   # collective (ncclCommInitRank)
    except Exception as e:                          # noqa: BLE001
        error = repr(e)
    errors, _ = _agree(d, "init", rank, world, error, b"", min(timeout, 120.0))
    if errors:
        if comm is not None:
            comm.close()
        _retire_after(d, ("ready", "init"), rank, world, timeout)
        raise TransportUnavailable("ncclCommInitRank failed on rank(s) %s" % errors)
    comm.barrier()                                  # everybody has read everything
    _retire(d, ("ready", "init"), rank, world)
    return comm


def _retire_after(directory, phases, rank, world, timeout):
    """On the failure paths there is no communicator to put a barrier behind the last read: one more file round does it."""
    try:
        _agree(directory, "bye", rank, world, None, b"", min(timeout, 60.0))
    except RuntimeError:                            # (TransportUnavailable is one)
        pass
    time.sleep(0.05)
    _retire(directory, tuple(phases) + ("bye",), rank, world)


def connect_or_fallback(rank=None, world=None, allow_file_fallback=False):
    """connect(); (comm, None) -- or, when RCCL cannot be brought up, (FileComm, reason) IF the caller
    allows it.  It should only when the ranks share devices (bench.py: fewer GPUs than ranks, the
    one-GPU smoke test of the multi-process path): on a node where every rank has its own GPU a
    multi-GPU run that silently finished on files would "pass" without ever touching xGMI, so there
    the failure is raised.  The decision is unanimous (see connect)."""
    try:
        return connect(rank, world), None
    except TransportUnavailable as e:
        if not allow_file_fallback:
            raise
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        return FileComm(rank, world), str(e)


def batch_seed0(rank, n_batches, pairs_per_batch, k):
    """First pair id (= synthetic seed) of batch k of `rank`: a rank owns n_batches consecutive blocks
    of pairs_per_batch pair ids (bench.py keeps two batches in flight)."""
    return int(pair_seeds(rank, n_batches * pairs_per_batch)[0]) + k * pairs_per_batch


def gathered_seed0s(world, n_batches, pairs_per_batch, k):
    """First pair id of every block of an all-gather of batch k's poses, in the order the gather
    returns them (rank order)."""
    return [batch_seed0(r, n_batches, pairs_per_batch, k) for r in range(world)]


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
