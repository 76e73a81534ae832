"""N > 1 path on CPU: world_size-2 gloo run of the pair sharding + pose gather
used by bench.py (tadataka_amd/sharding.py)."""
import os
import subprocess
import sys

import numpy as np

from conftest import REPO


def test_shard_bounds_cover_everything():
    from tadataka_amd.sharding import pair_seeds, shard_bounds
    for n in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert list(pair_seeds(3, 4)) == [12, 13, 14, 15]


def test_single_process_passthrough():
    from tadataka_amd.sharding import all_gather_poses, reduce_scalars
    p = np.arange(24.).reshape(2, 12)
    assert np.array_equal(all_gather_poses(p), p)
    assert np.array_equal(reduce_scalars([1., 2.], "sum"), [1., 2.])


def test_two_rank_gloo_shard_and_gather(tmp_path):
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29613",
           os.path.join(REPO, "tests", "_dist_worker.py"), out, "3"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    got = np.load(out)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _dist_worker import estimate_pair
    expected = np.array([estimate_pair(s) for s in range(6)])      # single process, all pairs
    assert got["world"] == 2
    assert got["gathered"].shape == (6, 12)
    assert np.array_equal(got["gathered"], expected)               # rank order, nothing lost
    assert np.array_equal(got["stats"], [2., 3.]) and got["total"][0] == 6.


def test_unique_id_rendezvous_without_gpu(tmp_path):
    """sharding.connect(): rank 0 publishes the 128-byte id in a file keyed by MASTER_PORT and the
    launcher's pid, the other ranks wait for it, rank 0 removes it -- with a recording stand-in for
    the RCCL communicator (three ranks, started out of order)."""
    import glob
    import time
    worker = os.path.join(REPO, "tests", "_dist_worker.py")
    procs, outs = [], []
    for rank in (2, 1, 0):                                        # rank 0 last: the others must wait
        out = str(tmp_path / f"uid{rank}.npy")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29714", TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, worker, "--rendezvous", out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        if rank != 0:
            time.sleep(0.2)
    for p in procs:
        stdout, _ = p.communicate(timeout=120)
        assert p.returncode == 0, stdout[-2000:]
    uids = [np.load(o) for o in outs]
    assert uids[0].shape == (128,) and all(np.array_equal(u, uids[0]) for u in uids)
    assert glob.glob(str(tmp_path / "tdk_rccl_*")) == []          # rank 0 cleaned up


def test_file_comm_fallback_three_ranks(tmp_path):
    """FileComm (what bench.py falls back to when RCCL cannot be initialised): gather / reduce /
    barrier across three processes."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from tadataka_amd import sharding
rank = int(os.environ["RANK"])
c = sharding.FileComm(rank, 3, "t")
g = c.all_gather(np.full((2, 12), float(rank)))
assert g.shape == (6, 12) and np.array_equal(g[:, 0], [0, 0, 1, 1, 2, 2])
assert np.array_equal(c.all_reduce([rank, 1.0], "sum"), [3.0, 3.0])
assert np.array_equal(c.all_reduce([rank, 1.0], "max"), [2.0, 1.0])
for _ in range(5):
    c.barrier()
pg = sharding.PoseGather(2, c)
pg.start(np.full((2, 12), 10.0 + rank))
assert np.array_equal(pg.finish()[:, 0], [10, 10, 11, 11, 12, 12])
''' % REPO
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, RANK=str(r), TMPDIR=str(tmp_path)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out[-2000:]
