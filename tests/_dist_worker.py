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


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--rendezvous":
    rendezvous_main()
    sys.exit(0)


if __name__ == "__main__":
    main()
