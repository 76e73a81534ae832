#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: "warp+residual+J^T J
Mpixels/sec per DVO iter; frame-pairs/sec at 1/2/4/8 GPUs").

A *step* is one full DVO pose estimation (PoseChangeEstimator: 3-level pyramid,
ratio 1.5, max_iter 20, weights="huber" -- BASELINE configs[1]) over one batch of
independent synthetic 640x480 frame pairs resident in HBM: build the pyramids,
then per level the fused evaluate/solve Gauss-Newton loop, all pairs in lock
step on the device.  With `--double-buffer` two batches alternate: every step
builds the pyramid of the batch the NEXT step will estimate -- queued on that
batch's own stream, underneath the estimation of the current batch -- and
estimates the current one (one build + one estimation per step either way;
measured +3 %, the estimation already saturates the chip's FP64 issue/power).  `value` counts every source pixel pushed through one DVO
iteration (= one calc_pose_update + one photometric_error at one pose, which
the fused kernel does in a single pass): sum over levels, iterations and still
running pairs of the level's pixel count, divided by wall time.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns
its own shard of `--pairs` pairs (weak scaling; no data-path collective inside
the estimation) and the recovered poses are all-gathered over RCCL at the end
of each step.

One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PX_EVAL = 24.0  # fused evaluation reads D0, I0, I1 once (f64); see DESIGN.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=256, help="frame pairs per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--weights", default="huber", choices=["none", "huber", "student-t", "tukey"])
    ap.add_argument("--double-buffer", action="store_true",
                    help="two batches: the next batch's pyramid is built under the current estimation")
    ap.add_argument("--anti-aliasing", action="store_true",
                    help="pyramid with skimage's Gaussian prefilter instead of SURVEY cfg2's plain bilinear rescale")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def true_poses(n, seed0):
    from tadataka_amd import synthetic
    out = np.empty((n, 12))
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        omega, t = synthetic.random_pose(rng)
        out[i, :9] = synthetic.rodrigues(omega).ravel()
        out[i, 9:] = t
    return out


def cpu_baseline(batch, cam, seconds):
    """The CPU oracle (plain-C restatement, 1 thread) timed on this box's host
    cores on a bounded sample of the same workload: one 640x480 pair taken from
    the device batch, repeated DVO iterations (calc_pose_update with Huber
    weights + photometric_error) at full resolution."""
    from oracle import oracle as orc
    I0 = batch.download(0, 0, "I0"); D0 = batch.download(0, 0, "D0"); I1 = batch.download(0, 0, "I1")
    GX, GY = orc.image_gradient(I1)
    R, t, T = np.eye(3), np.zeros(3), np.eye(4)
    orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, R, t, "huber")   # warm-up
    n_iter, t0 = 0, time.perf_counter()
    while True:
        orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, R, t, "huber")
        orc.photometric_error_sums(I0, D0, I1, cam, cam, T)
        n_iter += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": I0.size * n_iter / dt / 1e6, "unit": "Mpx/s", "cores": 1, "kind": "port",
            "sample": f"1 pair {I0.shape[1]}x{I0.shape[0]}, {n_iter} DVO iterations "
                      f"(calc_pose_update huber + photometric_error) in {dt:.1f} s, oracle/tdk_oracle.c -O2",
            "host_cpu": _cpu_model(), "host_cores_total": os.cpu_count()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = args.gpus > 1 or world > 1

    # torch (only needed for torch.distributed / RCCL when N > 1) bundles its own
    # libamdhip64 with the same soname as /opt/rocm's: whichever is loaded first
    # serves the whole process, so bring torch up BEFORE libtadataka_hip.so.
    dist = torch = None
    if distributed:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world, rank = dist.get_world_size(), dist.get_rank()

    from tadataka_amd import _lib, ops, sharding, synthetic
    _lib.require_gpu()
    _lib.call("tdk_set_device", local_rank)

    B, H, W = args.pairs, args.height, args.width
    cam = synthetic.camera_for(W, H)
    mode = ops.WEIGHT_MODES[None if args.weights == "none" else args.weights]
    n_batches = 2 if args.double_buffer else 1
    # this rank's shard of the pair ids: n_batches consecutive blocks of B pairs
    seeds = [int(sharding.pair_seeds(rank, n_batches * B)[0]) + k * B for k in range(n_batches)]
    batches = []
    for seed0 in seeds:
        bt = ops.DvoBatch(B, H, W, n_levels=args.levels, ratio=1.5)
        bt.fill_synthetic(cam, true_poses(B, seed0), seed0=seed0, noise=0.02)
        bt.set_anti_aliasing(args.anti_aliasing)
        batches.append(bt)
    batch = batches[0]
    ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    device = torch.device("cuda", local_rank) if distributed else None
    counter = [0]
    gather = sharding.PoseGather(B, dist, device)
    batches[0].build_pyramid()

    def step():
        k = counter[0]
        counter[0] += 1
        cur = batches[k % n_batches]
        if n_batches == 1:
            cur.build_pyramid()
        else:
            batches[(k + 1) % n_batches].build_pyramid()   # asynchronous, on that batch's stream
        poses, px = cur.estimate(cam, cam, ident, mode, args.max_iter)
        # the only exchange: the recovered poses, all-gathered (RCCL over xGMI).  The gather of
        # this step is queued now and collected after the next step's estimation (flush() below
        # collects the last one inside the timed region).
        previous = gather.finish() if gather.pending else None
        gather.start(poses)
        return previous, px, seeds[k % n_batches]

    def flush():
        return gather.finish()

    def fence():
        _lib.call("tdk_sync")
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if gather.pending:
        flush()
    for bt in batches:
        bt.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    pixels = 0
    for _ in range(args.steps):
        _, px, last_seed = step()
        pixels += px
    poses = flush() if gather.pending else None   # all-gathered poses of the last step
    fence()
    elapsed = time.perf_counter() - t0
    prof = {"launches": 0, "total_ms": 0.0, "pixels": 0}
    for bt in batches:
        for key, val in bt.get_profile().items():
            prof[key] += val
        bt.set_profiling(False)

    elapsed = float(sharding.reduce_scalars([elapsed], "max", dist, device)[0])
    pixels_all = float(sharding.reduce_scalars([float(pixels)], "sum", dist, device)[0])

    if rank == 0:
        # rank r's batch of the last step holds pairs [r * n_batches * B + offset, ... + B)
        offset = last_seed - seeds[0]
        truth = np.concatenate([true_poses(B, r * n_batches * B + offset) for r in range(world)])
        assert poses.shape == truth.shape
        t_err = float(np.max(np.linalg.norm(poses[:, 9:] - truth[:, 9:], axis=1)))
        kernel_ms = prof["total_ms"] / max(prof["launches"], 1)
        bytes_per_launch = BYTES_PER_PX_EVAL * prof["pixels"] / max(prof["launches"], 1)
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        pmc_file = os.path.join(REPO, "profiles", "pmc_dvo_eval.json")
        if os.path.exists(pmc_file):
            try:
                traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
            except (ValueError, OSError):
                traffic = None
        out = {
            "metric": "warp+residual+JtJ Mpixels/sec per DVO iter",
            "value": pixels_all / elapsed / 1e6,
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "DVO pose estimation (PoseChangeEstimator), batch of independent "
                                   f"{W}x{H} frame pairs, {args.levels}-level pyramid ratio 1.5, "
                                   f"weights={args.weights}, max_iter={args.max_iter}",
                       "pairs_per_gpu": B, "batches_in_flight": n_batches,
                       "pyramid": "anti-aliased (gaussian prefilter + bilinear)" if args.anti_aliasing else "bilinear",
                       "height": H, "width": W, "levels": args.levels,
                       "weights": args.weights, "max_iter": args.max_iter,
                       "parallelism": f"pair-shard x{world}" if world > 1 else "single GPU"},
            "frame_pairs_per_s": B * world * args.steps / elapsed,
            "dvo_iterations_per_pair_per_step": pixels / args.steps / B / (H * W),
            "max_translation_error": t_err,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"k_dvo_eval<{args.weights}> (full-resolution level)",
                         "bytes_per_px": BYTES_PER_PX_EVAL,
                         "px_per_launch": prof["pixels"] / max(prof["launches"], 1),
                         "kernel_ms": kernel_ms, "launches": prof["launches"],
                         "limiter": "FP64 issue at the 1400 W package power cap (DESIGN.md 5.1), not HBM"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch, cam, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    for bt in batches:
        bt.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
