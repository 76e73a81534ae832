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


def _spawn_ranks(tmp_path, world, port, mode, extra_env=None, order=None):
    worker = os.path.join(REPO, "tests", "_dist_worker.py")
    procs, outs = [], []
    import time
    for rank in (order or range(world)):
        out = str(tmp_path / f"out{rank}.npy")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TMPDIR=str(tmp_path), TDK_TEST_MODE=mode, **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, worker, "--rendezvous", out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        if order:
            time.sleep(0.2)
    for p in procs:
        stdout, _ = p.communicate(timeout=180)
        assert p.returncode == 0, stdout[-3000:]
    return outs


def test_unique_id_rendezvous_without_gpu(tmp_path):
    """sharding.connect(): the ranks agree through a private directory of the launch that all of them
    can open RCCL (rank 0's message carries the 128-byte id), initialise, agree again, and clean up --
    with a recording stand-in for the RCCL communicator (three ranks, started out of order)."""
    import glob
    outs = _spawn_ranks(tmp_path, 3, 29714, "connect", order=(2, 1, 0))     # rank 0 last: the others must wait
    uids = [np.load(o) for o in outs]
    assert uids[0].shape == (128,) and all(np.array_equal(u, uids[0]) for u in uids)
    assert glob.glob(str(tmp_path / "tdk_rdv_*")) == []           # the last one out removed the directory


def test_one_rank_without_rccl_fails_every_rank_together(tmp_path):
    """A rank that cannot open librccl says so BEFORE anybody enters ncclCommInitRank: every rank raises
    TransportUnavailable with the reason (no hang, no mixed transports); where the caller allows the file
    fallback (ranks sharing a GPU) every rank gets it."""
    import glob
    _spawn_ranks(tmp_path, 3, 29715, "must_raise", {"TDK_TEST_FAIL_RANK": "1"})
    outs = _spawn_ranks(tmp_path, 3, 29716, "fallback", {"TDK_TEST_FAIL_RANK": "0"})   # rank 0 itself fails
    assert all(np.array_equal(np.load(o)[:, 0], [0, 1, 2]) for o in outs)
    assert glob.glob(str(tmp_path / "tdk_rdv_*")) == []


def test_launch_key_and_stale_directories(tmp_path, monkeypatch):
    """The rendezvous directory is private (0700), keyed by port + launcher pid + launcher start time (a
    crashed run that left files behind under a reused pid and port is another directory), and
    TDK_RENDEZVOUS_KEY overrides it for every transport."""
    import stat
    from tadataka_amd import sharding
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29000")
    monkeypatch.delenv("TDK_RENDEZVOUS_KEY", raising=False)
    key = sharding.launch_key()
    assert key.startswith("29000_%d_" % os.getppid()) and key.split("_")[2] not in ("", "0")
    d = sharding.rendezvous_dir()
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    sharding._publish(d, "ready_0", b"ok:x")
    assert stat.S_IMODE(os.stat(os.path.join(d, "ready_0")).st_mode) == 0o600
    monkeypatch.setenv("TDK_RENDEZVOUS_KEY", "my/key 1")
    assert sharding.launch_key() == "my_key_1" and sharding.rendezvous_dir().endswith("tdk_rdv_my_key_1")
    c = sharding.FileComm(0, 1)
    assert "tdk_rdv_my_key_1" in c._dir
    c.close()
    assert not os.path.exists(c._dir)


def test_file_comm_fallback_three_ranks(tmp_path):
    """FileComm (what bench.py may fall back to when RCCL cannot be initialised and ranks share a GPU):
    gather / reduce / barrier across three processes, directory removed at the end."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from tadataka_amd import sharding
rank = int(os.environ["RANK"])
c = sharding.FileComm(rank, 3)
g = c.all_gather(np.full((2, 12), float(rank)))
assert g.shape == (6, 12) and np.array_equal(g[:, 0], [0, 0, 1, 1, 2, 2])
assert np.array_equal(c.all_reduce([rank, 1.0], "sum"), [3.0, 3.0])
assert np.array_equal(c.all_reduce([rank, 1.0], "max"), [2.0, 1.0])
for _ in range(5):
    c.barrier()
pg = sharding.PoseGather(2, c)
pg.start(np.full((2, 12), 10.0 + rank))
assert np.array_equal(pg.finish()[:, 0], [10, 10, 11, 11, 12, 12])
c.close()
''' % REPO
    import glob
    procs = [subprocess.Popen([sys.executable, "-c", code],
                              env=dict(os.environ, RANK=str(r), TMPDIR=str(tmp_path), TDK_RENDEZVOUS_KEY="fc3"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out[-2000:]
    assert glob.glob(str(tmp_path / "tdk_rdv_fc3" / "filecomm" / "*.npy")) == []


def test_bench_pair_bookkeeping_world_8(tmp_path):
    """bench.py's sharding arithmetic end to end over eight real processes: every rank owns two
    consecutive blocks of pair ids (two batches in flight), the gather of step k is collected after step
    k + 1, and rank 0 compares the gathered poses with the blocks `gathered_seed0s` names, in rank order."""
    outs = _spawn_ranks(tmp_path, 8, 29717, "bench_order", {"TDK_RENDEZVOUS_KEY": "w8"})
    last = np.load(outs[0])
    assert last.shape == (24, 12)
    assert np.array_equal(last[:, 0], np.concatenate([np.arange(6 * r, 6 * r + 3) for r in range(8)]))
    assert all(np.array_equal(np.load(o), last) for o in outs)


def test_rccl_comm_through_the_c_abi_signatures_world_8(tmp_path):
    """Eight processes run the product's sharding.connect() -> RcclComm -> PoseGather (device-resident gather path)
    against a stand-in whose entries are ctypes callbacks with the prototypes of include/tadataka_hip.h and gloo
    behind them (tests/_dist_worker.py: GlooBackedAbi): rendezvous with the unique id, ncclCommInitRank's place,
    bench.py's order of collectives, error by status, destroy.  Leaves csrc/comm.hip as the only code of the N > 1
    path that has not run with real peers."""
    import glob
    world = 8
    worker = os.path.join(REPO, "tests", "_dist_worker.py")
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / f"abi{rank}.npy")
        outs.append(out)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29731", TMPDIR=str(tmp_path), OMP_NUM_THREADS="1", TDK_RENDEZVOUS_KEY="abi8")
        procs.append(subprocess.Popen([sys.executable, worker, "--abi", out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        stdout, _ = p.communicate(timeout=300)
        assert p.returncode == 0, stdout[-3000:]
    last = np.load(outs[0])
    assert last.shape == (world * 3, 12)
    assert np.array_equal(last[:, 0], np.concatenate([np.arange(6 * r, 6 * r + 3) for r in range(world)]) + 4000.0)
    assert all(np.array_equal(np.load(o), last) for o in outs)
    assert glob.glob(str(tmp_path / "tdk_rdv_*")) == []


def test_bench_spawner_names_every_failed_rank(tmp_path):
    """`python bench.py --dry-ranks 2` on a box WITHOUT a GPU: both workers fail (there is no CPU fallback) and the
    spawner says, per rank, exit code, device rule and the last lines of that rank's stderr -- not exit codes only."""
    from conftest import _has_gpu
    import pytest
    if _has_gpu():
        pytest.skip("needs a box without a GPU: the workers must fail")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-ranks", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(os.environ, TMPDIR=str(tmp_path)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""
    for r in (0, 1):
        assert f"---- rank {r}: exit code 1, LOCAL_RANK={r}" in p.stderr
    assert p.stderr.count("no MI355X / HIP device visible") >= 2
    assert "bench worker exit codes: [1, 1]" in p.stderr
