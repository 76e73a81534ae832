"""Worker for tests/test_distributed_cpu.py (gloo, world_size 2, CPU): each rank
estimates the poses of its own shard of frame pairs -- with the CPU oracle
standing in for the GPU, this is a test -- and the shards are gathered exactly
as bench.py does it."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def estimate_pair(seed, h=32, w=40):
    from scipy.spatial.transform import Rotation
    from oracle import oracle as orc
    from tadataka_amd import synthetic
    pair = synthetic.make_pair(h, w, seed=int(seed))
    rot, t = orc.dvo_estimate_level(pair["I0"], pair["D0"], pair["I1"], pair["cam"], pair["cam"],
                                    Rotation.from_rotvec(np.zeros(3)), np.zeros(3), "huber", 20)
    return np.concatenate([rot.as_matrix().ravel(), t])


class GlooComm(object):
    """The communicator interface of tadataka_amd.sharding on torch.distributed/gloo:
    stands in for RCCL where there is no GPU (test infrastructure only)."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_gather(self, array):
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64))
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return torch.cat(parts, dim=0).numpy()

    def all_reduce(self, values, op):
        import torch
        t = torch.from_numpy(np.array(values, dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()


def main():
    import torch.distributed as dist
    from tadataka_amd import sharding
    out_path, pairs_per_rank = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    comm = GlooComm(dist)
    rank, world = comm.rank, comm.world
    seeds = sharding.pair_seeds(rank, pairs_per_rank)
    local = np.array([estimate_pair(s) for s in seeds])
    gathered = sharding.all_gather_poses(local, comm)
    # the pipelined form bench.py uses: queue the gather of "step k", collect it after "step k + 1"
    pg = sharding.PoseGather(pairs_per_rank, comm)
    pg.start(local)
    first = pg.finish()
    pg.start(local + 1.0)
    second = pg.finish()
    assert np.array_equal(first, gathered) and np.array_equal(second, gathered + 1.0)
    stats = sharding.reduce_scalars([float(rank + 1), float(len(seeds))], "max", comm)
    total = sharding.reduce_scalars([float(len(seeds))], "sum", comm)
    comm.barrier()
    if rank == 0:
        np.savez(out_path, gathered=gathered, stats=stats, total=total, world=world)
    dist.destroy_process_group()




def rendezvous_main():
    """`python _dist_worker.py --rendezvous out.npy`: sharding.connect() with the RCCL communicator
    replaced by a recorder -- exercises the unique-id exchange (file keyed by MASTER_PORT and the
    launcher's pid) without a GPU."""
    from tadataka_amd import sharding

    class FakeComm(object):
        def __init__(self, rank, world, unique_id):
            self.rank, self.world, self.uid = rank, world, bytes(unique_id)

        @staticmethod
        def available():
            if os.environ.get("TDK_TEST_FAIL_RANK") == os.environ["RANK"]:
                raise RuntimeError("cannot open librccl.so (simulated)")

        @staticmethod
        def unique_id():
            return bytes(((np.arange(128) * 7 + os.getpid()) % 256).astype(np.uint8))

        def close(self):
            pass

        def barrier(self):              # every rank drops a marker, then waits for all of them
            import time
            d = os.environ["TMPDIR"]
            open(os.path.join(d, "arrived_%d" % self.rank), "w").close()
            t0 = time.time()
            while not all(os.path.exists(os.path.join(d, "arrived_%d" % r)) for r in range(self.world)):
                assert time.time() - t0 < 60.0
                time.sleep(0.01)

    sharding.RcclComm = FakeComm
    mode = os.environ.get("TDK_TEST_MODE", "connect")
    if mode == "connect":
        comm = sharding.connect(timeout=60.0)
        np.save(sys.argv[2], np.frombuffer(comm.uid, dtype=np.uint8))
        assert comm.world == int(os.environ["WORLD_SIZE"]) and comm.rank == int(os.environ["RANK"])
    elif mode == "must_raise":          # one rank cannot open RCCL: EVERY rank raises, none hangs
        try:
            sharding.connect_or_fallback(allow_file_fallback=False)
        except sharding.TransportUnavailable as e:
            assert "simulated" in str(e)
            np.save(sys.argv[2], np.zeros(1))
        else:
            raise AssertionError("connect succeeded although a rank had no RCCL")
    elif mode == "fallback":            # ... or, where ranks share a device, every rank gets the file transport
        comm, why = sharding.connect_or_fallback(allow_file_fallback=True)
        assert comm.kind == "file" and "simulated" in why
        got = comm.all_gather(np.full((1, 12), float(comm.rank)))
        assert np.array_equal(got[:, 0], np.arange(comm.world))
        comm.close()
        np.save(sys.argv[2], got)
    elif mode == "bench_order":         # bench.py's pair bookkeeping over a real multi-process gather
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        comm = sharding.FileComm(rank, world)
        B, n_batches = 3, 2
        pg = sharding.PoseGather(B, comm)
        for k in range(3):              # steps 0, 1, 2 use batches 0, 1, 0
            kb = k % n_batches
            s0 = sharding.batch_seed0(rank, n_batches, B, kb)
            poses = np.tile(np.arange(s0, s0 + B, dtype=np.float64)[:, None], (1, 12))   # "pose" = pair id
            previous = pg.finish() if pg.pending else None
            pg.start(poses)
            if previous is not None:
                want = np.concatenate([np.arange(x, x + B) for x in sharding.gathered_seed0s(world, n_batches, B, (k - 1) % n_batches)])
                assert np.array_equal(previous[:, 0], want), (k, previous[:, 0], want)
        last = pg.finish()
        want = np.concatenate([np.arange(x, x + B) for x in sharding.gathered_seed0s(world, n_batches, B, 0)])
        assert np.array_equal(last[:, 0], want)
        total = sharding.reduce_scalars([float(B)], "sum", comm)
        assert total[0] == world * B
        comm.close()
        np.save(sys.argv[2], last)


class GlooBackedAbi(object):
    """Stands in for libtadataka_hip.so's tdk_comm_* / tdk_dvo_gather_poses_* entries on a box without GPUs: every
    entry is a ctypes callback built from the SAME prototype tadataka_amd/_lib.py declares for the real symbol
    (include/tadataka_hip.h), so sharding.RcclComm -- the class the product uses, unchanged -- marshals its arguments
    through identical signatures, and the collective behind it is torch.distributed/gloo instead of RCCL.  What this
    leaves untested before the first multi-GPU run is csrc/comm.hip itself.  Test infrastructure only."""

    NAMES = ("tdk_comm_available", "tdk_comm_unique_id", "tdk_comm_create", "tdk_comm_destroy", "tdk_comm_rank",
             "tdk_comm_all_gather", "tdk_comm_all_reduce", "tdk_comm_barrier", "tdk_dvo_gather_poses_start",
             "tdk_dvo_gather_poses_finish")

    def __init__(self):
        import ctypes as C
        from tadataka_amd import _lib
        self.C, self.error = C, b""
        self.comms, self.batches, self.calls = {}, {}, []
        for name in self.NAMES:
            proto = C.CFUNCTYPE(C.c_int, *_lib.PROTOTYPES[name])
            setattr(self, name, proto(self._guard(name, getattr(self, "_" + name[4:]))))

    def _guard(self, name, fn):
        def entry(*args):
            self.calls.append(name)
            try:
                fn(*args)
                return 0
            except Exception as e:                  # noqa: BLE001  (a status + tdk_last_error, like the library)
                self.error = ("%s: %r" % (name, e)).encode()
                return -2
        return entry

    def tdk_last_error(self):
        return self.error

    def _arr(self, ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(int(n),))

    def _comm_available(self):
        if os.environ.get("TDK_TEST_FAIL_RANK") == os.environ["RANK"]:
            raise RuntimeError("cannot open librccl.so (simulated)")

    def _comm_unique_id(self, id128):
        self._arr(id128, 128)[:] = (np.arange(128) * 11 + os.getpid()) % 256

    def _comm_create(self, id128, rank, world, out):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=int(rank), world_size=int(world))
        mine = torch.from_numpy(self._arr(id128, 128).copy())
        parts = [torch.empty_like(mine) for _ in range(int(world))]
        dist.all_gather(parts, mine)
        if not all(torch.equal(q, parts[0]) for q in parts):
            raise RuntimeError("ranks were created with different unique ids")
        handle = 0x1000 + len(self.comms)
        self.comms[handle] = {"rank": int(rank), "world": int(world), "pending": None}
        out[0] = handle

    def _comm_destroy(self, h):
        del self.comms[h]

    def _comm_rank(self, h, rank, world):
        rank[0], world[0] = self.comms[h]["rank"], self.comms[h]["world"]

    def _gather(self, h, a):
        import torch
        import torch.distributed as dist
        mine = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        parts = [torch.empty_like(mine) for _ in range(self.comms[h]["world"])]
        dist.all_gather(parts, mine)
        return torch.cat(parts).numpy()

    def _comm_all_gather(self, h, send, count, recv):
        self._arr(recv, self.comms[h]["world"] * count)[:] = self._gather(h, self._arr(send, count))

    def _comm_all_reduce(self, h, values, count, op):
        import torch
        import torch.distributed as dist
        if op not in (0, 1):
            raise ValueError("op: 0 sum, 1 max")
        v = self._arr(values, count)
        t = torch.from_numpy(v.copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        v[:] = t.numpy()

    def _comm_barrier(self, h):
        import torch.distributed as dist
        assert h in self.comms
        dist.barrier()

    def _dvo_gather_poses_start(self, batch, h):
        if self.comms[h]["pending"] is not None:
            raise RuntimeError("one gather in flight per communicator")
        self.comms[h]["pending"] = self._gather(h, self.batches[batch].reshape(-1))

    def _dvo_gather_poses_finish(self, h, poses_all):
        got = self.comms[h]["pending"]
        if got is None:
            raise RuntimeError("no gather in flight")
        self._arr(poses_all, got.size)[:] = got
        self.comms[h]["pending"] = None


def abi_main():
    """`python _dist_worker.py --abi out.npy` under WORLD_SIZE ranks: the product's own sharding.connect() /
    RcclComm / PoseGather against GlooBackedAbi, in bench.py's order of calls (two batches in flight, the gather of
    step k collected after step k + 1, the MAX / SUM reductions, the one-hot device report, barrier, close)."""
    import ctypes as C
    from tadataka_amd import _lib, sharding
    fake = GlooBackedAbi()
    _lib._lib = fake                                     # what _lib.load() returns from now on
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    comm, why = sharding.connect_or_fallback(allow_file_fallback=False)
    assert why is None and comm.kind == "rccl" and (comm.rank, comm.world) == (rank, world)

    class Batch(object):                                 # what PoseGather needs of a DvoBatch
        def __init__(self, handle, n_pairs):
            self._h, self.n_pairs = C.c_void_p(handle), n_pairs

    B, n_batches = 3, 2
    batches = [Batch(0x2000 + k, B) for k in range(n_batches)]
    pg = sharding.PoseGather(B, comm)
    for k in range(5):                                   # steps 0 .. 4 use batches 0, 1, 0, 1, 0
        kb = k % n_batches
        s0 = sharding.batch_seed0(rank, n_batches, B, kb)
        poses = np.tile(np.arange(s0, s0 + B, dtype=np.float64)[:, None], (1, 12)) + 1000.0 * k
        fake.batches[batches[kb]._h.value] = poses       # "where the device loop left them"
        previous = pg.finish() if pg.pending else None
        pg.start(poses, batches[kb])
        if previous is not None:
            want = np.concatenate([np.arange(x, x + B) for x in
                                   sharding.gathered_seed0s(world, n_batches, B, (k - 1) % n_batches)]) + 1000.0 * (k - 1)
            assert previous.shape == (world * B, 12) and np.array_equal(previous[:, 0], want), (k, previous[:, 0], want)
    last = pg.finish()
    dt = float(sharding.reduce_scalars([1.0 + rank], "max", comm)[0])
    assert dt == float(world)
    onehot = np.zeros(world)
    onehot[rank] = float(rank % 8)
    devices = [int(v) for v in sharding.reduce_scalars(onehot, "sum", comm)]
    assert devices == [r % 8 for r in range(world)]
    host = comm.all_gather(np.full((2, 12), float(rank)))          # the host-buffer collective
    assert host.shape == (2 * world, 12) and np.array_equal(host[::2, 0], np.arange(world))
    try:                                                 # a second gather in flight is an error of the ABI, by status
        comm.gather_poses_start(batches[0])
        comm.gather_poses_start(batches[0])
    except _lib.TdkError as e:
        assert "one gather in flight" in str(e)
    else:
        raise AssertionError("two gathers in flight were accepted")
    comm.gather_poses_finish()
    comm.barrier()
    comm.close()
    assert "tdk_comm_destroy" in fake.calls and fake.comms == {}
    np.save(sys.argv[2], last)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--abi":
    abi_main()
    sys.exit(0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--rendezvous":
    rendezvous_main()
    sys.exit(0)


if __name__ == "__main__":
    main()
