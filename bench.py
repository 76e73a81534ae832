#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: "warp+residual+J^T J
Mpixels/sec per DVO iter; frame-pairs/sec at 1/2/4/8 GPUs") and, on one GPU, every
other BASELINE config as an entry of `workloads` in the same JSON line.

Headline.  A *step* is one full DVO pose estimation (PoseChangeEstimator: 3-level
pyramid, ratio 1.5, max_iter 20, weights="huber" -- BASELINE configs[1]) over one
batch of independent synthetic 640x480 frame pairs resident in HBM: build the
pyramids (anti-aliased, as skimage.rescale does by default -- the same constant
the drop-in tadataka.vo.dvo uses), then per level the fused evaluate/solve
Gauss-Newton loop, all pairs in lock step on the device.  `value` is SURVEY 8(d)'s
unit: one DVO iter = one calc_pose_update (warp, mask, gradients, Jacobian, weights,
27-entry reduction, solve) + one photometric_error over the source pixels of a level;
`value` = sum over levels and pairs of (level pixels x pose updates solved) / wall time.
The reference runs n updates and n + 1 errors per level and pair; both are counted on
the device (tdk_dvo_get_counts) and reported separately (`update_mpx_per_s`,
`error_mpx_per_s`, `updates_per_step`, `errors_per_step`); round 1-2's number (every
PhotometricError evaluation counted as an iteration) stays under
`error_evaluations_mpx_per_s`.  `frame_pairs_per_s` is the unambiguous companion.  Pair 0
of the batch is the seed-0 pair of tests/golden/dvo_vga_pyramid.npz: its pose is
checked against what the REFERENCE's own PoseChangeEstimator returned.

N > 1: one process per GPU (launched by torch.distributed.run as the driver does,
or by this script itself when WORLD_SIZE is unset), every rank owns its own shard
of `--pairs` pairs (weak scaling; no data-path collective inside the estimation)
and the recovered poses are all-gathered with RCCL through the C ABI
(tdk_comm_*) at the end of each step.  No PyTorch in any process.

Timing: W warm-up steps, then blocks of exactly K steps, each bracketed by a
barrier + device synchronisation on both sides and reduced with MAX over ranks;
blocks repeat until --min-seconds of timed work have run (so that a GPU-busy
sampler sees the run) and ms_per_step is the mean over all timed steps.

One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (MI355X_MICROARCH.md; SURVEY 8(d)'s secondary ceiling)
# Algorithmic bytes per unit
BYTES_PER_PX_EVAL = 24.0      # fused evaluation reads D0, I0, I1 once (f64)
BYTES_PER_PX_WARP = 56.0      # increment_age 24 + propagate 32
BYTES_PER_OBS_BA = 56.0       # point 24 + x_true 16 + two int64 indices 16
SD_PARAMS = (0.5, 10.0, 0.01, 0.01, 0.002, 0.02)
SD_DEFAULTS = (1.0, 10.0, 0.01)


def parse_args():
    import tadataka_amd
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
    ap.add_argument("--single-buffer", dest="double_buffer", action="store_false",
                    help="one batch: its pyramid is rebuilt and then estimated, strictly one after the other.  Default: "
                         "two batches in flight per GPU -- every step still builds one pyramid and estimates one batch, "
                         "but the NEXT batch's pyramid is queued on that batch's own stream and runs under the current "
                         "batch's estimation")
    ap.set_defaults(double_buffer=True)
    ap.add_argument("--pyramid", choices=["skimage", "skimage-depth-level0", "ideal", "bilinear"], default="skimage",
                    help="skimage = every level as skimage.transform.rescale returns it, to the bit, level 0 through "
                         "rescale(., 1.0) and clip=True included -- what the reference builds (for 640x480 x 3 levels with "
                         "the plans of tests/golden/skimage_dvo.npz, so that pair 0 can be held against the reference "
                         "run on the real scikit-image; else with this interpreter's own plans); skimage-depth-level0 = "
                         "the same, but only the depth map gets a level 0 of its own (the images' level 0 is the frame: "
                         "1e-13 from skimage's, poses unchanged at 1e-16); ideal = ideal sample positions, libm "
                         "kernels, level 0 = the frame, no clip (rounds 1-4); bilinear = ideal without the prefilter")
    ap.add_argument("--min-seconds", type=float, default=10.0,
                    help="repeat the timed block of --steps steps until this much timed work has run (default 10 s: a "
                         "GPU-busy sampler with a 5 s period cannot miss it; the headline is the LAST thing the run does)")
    ap.add_argument("--config", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2 (default, the headline): BASELINE configs[1], 640x480, 3 levels, max_iter 20.  cfg4: one "
                         "GPU's shard of BASELINE configs[3] -- 64 pairs of 1280x720 per GPU, 1 level, 2 iterations -- "
                         "so that a multi-GPU run can be quoted on the workload BASELINE names for it")
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="N worker processes that SHARE the visible GPU(s) and exchange through files instead of RCCL: "
                         "the multi-rank bookkeeping of this script (sharded seeds, two batches in flight, the gather "
                         "of step k collected after step k + 1, MAX / SUM reductions) runs end to end on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-workloads", action="store_true", help="headline only")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="skip the two child runs under rocprofv3 --pmc that measure roofline.traffic in this run")
    ap.add_argument("--no-solo-pass", action="store_true",
                    help="skip the untimed single-batch pass behind roofline.by_level / pyramid_roofline (profiling runs)")
    args = ap.parse_args()
    if args.config == "cfg4":           # explicit --pairs / --height / ... still win (reduced smoke runs)
        given = set(a.split("=")[0] for a in sys.argv[1:] if a.startswith("--"))
        for flag, name, value in (("--pairs", "pairs", 64), ("--height", "height", 720), ("--width", "width", 1280),
                                  ("--levels", "levels", 1), ("--max-iter", "max_iter", 2)):
            if flag not in given:
                setattr(args, name, value)
    return args


def true_poses(n, seed0):
    from tadataka_amd import synthetic
    out = np.empty((n, 12))
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        omega, t = synthetic.random_pose(rng)
        out[i, :9] = synthetic.rodrigues(omega).ravel()
        out[i, 9:] = t
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed_loop(fn, seconds, min_iter=2):
    fn()                                            # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= min_iter:
            return n, dt


def eval_profile(batch):
    """Every full-resolution k_dvo_eval launch since set_profiling(True): full evaluations, probes, mixed."""
    tot = {"launches": 0, "total_ms": 0.0, "pixels": 0}
    for kind in ("full", "probe", "mixed"):
        for key, val in batch.get_profile(kind).items():
            tot[key] += val
    return tot


def roofline(bytes_per_launch, kernel_ms, **extra):
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kernel_ms}
    out.update(extra)
    return out


def load_profile_json(name):
    path = os.path.join(REPO, "profiles", name)
    try:
        return json.load(open(path))
    except (ValueError, OSError):
        return None


def measure_traffic_in_run(argv):
    """HBM bytes per full-resolution evaluation launch, measured NOW: two child runs of this script (headline
    only, a few steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, KiB units,
    read side doubled on gfx950, exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes (and as
    profiles/summarize.py does for the committed profile).  Returns (dict | None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k in os.environ for k in ("ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_SDK_TOOL_LIBRARIES")):
        return None, "this run is itself under a profiler"
    keep = []
    skip_next = False
    for a in argv:                       # the workload flags of this run; its length flags are replaced
        if skip_next:
            skip_next = False
            continue
        if a in ("--gpus", "--steps", "--warmup", "--min-seconds", "--cpu-seconds", "--dry-ranks"):
            skip_next = True
            continue
        if a.split("=")[0] in ("--gpus", "--steps", "--warmup", "--min-seconds", "--cpu-seconds", "--dry-ranks"):
            continue
        keep.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + keep + [
        "--no-cpu-baseline", "--no-workloads", "--no-solo-pass", "--no-traffic-pass", "--min-seconds", "0.2",
        "--steps", "6", "--warmup", "2"]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    raw = {}
    tmp = tempfile.mkdtemp(prefix="tdk_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            try:
                subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=120, check=True)
            except (subprocess.SubprocessError, OSError) as e:
                return None, "rocprofv3 --pmc %s failed: %r" % (ctr, e)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    if r["Counter_Name"] != ctr or "k_dvo_eval" not in name:
                        continue
                    per.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            if not per:
                return None, "no k_dvo_eval rows in the %s pass" % ctr
            top = max(per)               # the full-resolution launches
            raw[ctr] = (sum(per[top]) / len(per[top]), len(per[top]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_kib, n_f = raw["FETCH_SIZE"]
    write_kib, n_w = raw["WRITE_SIZE"]
    return ({"hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0, "FETCH_SIZE_KiB_raw": fetch_kib,
             "WRITE_SIZE_KiB_raw": write_kib, "launches_counted": [n_f, n_w], "seconds": time.perf_counter() - t0},
            "two child runs of this command (headline only, 6 steps) under rocprofv3 --pmc FETCH_SIZE / --pmc "
            "WRITE_SIZE during this run; the full-resolution k_dvo_eval launches (the dominant kernel, full "
            "evaluations); read side doubled (gfx950 FETCH_SIZE correction)")


def measure_kernel_traffic(child, groups):
    """HBM bytes per launch of the kernels named in `groups` ({label: (kernel-name substrings)}), summed per label
    over its kernels (each launched once per step of `child`): two runs of `child` under rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE, corrected as in measure_traffic_in_run.  Returns ({label: bytes}, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k in os.environ for k in ("ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_SDK_TOOL_LIBRARIES")):
        return None, "this run is itself under a profiler"
    env = dict(os.environ, TMPDIR="/tmp")
    tmp = tempfile.mkdtemp(prefix="tdk_pmc_", dir="/tmp")
    raw = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            try:
                subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=150, check=True)
            except (subprocess.SubprocessError, OSError) as e:
                return None, "rocprofv3 --pmc %s failed: %r" % (ctr, e)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr:
                        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            raw[ctr] = per
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for label, subs in groups.items():
        total = 0.0
        for sub in subs:
            f = [v for k, vals in raw["FETCH_SIZE"].items() if sub in k for v in vals]
            w = [v for k, vals in raw["WRITE_SIZE"].items() if sub in k for v in vals]
            if not f or not w:
                return None, "no %s rows in the counter passes" % sub
            total += (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0
        out[label] = total
    return out, ("two child runs (%s) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE during this run; per-launch "
                 "averages summed over the kernels of a step; read side doubled (gfx950 FETCH_SIZE correction)"
                 % " ".join(os.path.basename(c) for c in child[1:2]))


def roofline_fp64(prof_kind):
    """SURVEY 8(d)'s secondary ceiling: FP64 vector rate of the evaluation kernel.  FLOPs per pixel
    come from the instruction counters of a rocprofv3 --pmc pass over the same kernel
    (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64; profiles/r05_fp64_mix.json, made by tools/fp64_mix.sh),
    the kernel time from this run's HIP events."""
    mix = load_profile_json("r05_fp64_mix.json") or load_profile_json("r03_fp64_mix.json")
    if not mix:
        return None
    out = {"bound": "fp64-vector", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
           "source": "profiles/r05_fp64_mix.json (rocprofv3 --pmc, tools/fp64_mix.sh, the round-5 kernels); kernel time: HIP events of this run",
           "by_mode": {}}
    flops, ms = 0.0, 0.0
    for kind in ("full", "probe"):
        pk = prof_kind.get(kind)
        m = mix.get(kind)
        if not pk or not pk["launches"] or not m:
            continue
        f = m["flops_per_px"] * pk["pixels"]
        flops += f
        ms += pk["total_ms"]
        ach = f / (pk["total_ms"] * 1e-3) / 1e12
        out["by_mode"][kind] = {"flops_per_px": m["flops_per_px"], "fp64_insts_per_px": m.get("fp64_insts_per_px"),
                                "achieved": ach, "frac": ach / FP64_VECTOR_PEAK_TFLOPS}
    if ms > 0:
        out["achieved"] = flops / (ms * 1e-3) / 1e12
        out["frac"] = out["achieved"] / FP64_VECTOR_PEAK_TFLOPS
    clock = mix.get("effective_clock_ghz")
    if clock:
        out["effective_clock_ghz_in_profile"] = clock
    return out


# ---------------------------------------------------------------------------
# CPU baselines (the oracle as the thing that is TIMED, on this box's host cores)
# ---------------------------------------------------------------------------
def dvo_cpu_baselines(I0, D0, I1, cam, seconds):
    """One 640x480 pair, repeated DVO iterations (calc_pose_update with Huber weights +
    photometric_error) at full resolution, single thread, three ways."""
    from oracle import numpy_port as npp
    from oracle import oracle as orc
    out = {}
    common = {"unit": "Mpx/s", "cores": 1, "host_cpu": _cpu_model(), "host_cores_total": os.cpu_count()}
    R, t, T = np.eye(3), np.zeros(3), np.eye(4)

    def c_iter():
        orc.dvo_normal_equations(I0, D0, I1, GX, GY, cam, cam, R, t, "huber")
        orc.photometric_error_sums(I0, D0, I1, cam, cam, T)

    for name, native, share in (("c_port_O3_native", True, 0.3), ("c_port_O2_exact", False, 0.15)):
        try:
            orc.use_library(orc.build_native() if native else None)
            GX, GY = orc.image_gradient(I1)
            n, dt = _timed_loop(c_iter, seconds * share)
            out[name] = dict(common, value=I0.size * n / dt / 1e6, kind="port",
                             sample=f"1 pair {I0.shape[1]}x{I0.shape[0]}, {n} DVO iterations (calc_pose_update "
                                    f"huber + photometric_error, full resolution, no pyramid work) in {dt:.1f} s; "
                                    "oracle/tdk_oracle.c " +
                                    ("gcc -O3 -march=native" if native else "gcc -O2 -ffp-contract=off (the exact checker build)"))
        except Exception as e:                      # noqa: BLE001  (a missing compiler must not kill the bench)
            out[name] = {"error": repr(e)}
        finally:
            orc.use_library(None)
    # the same C restatement on every host core "for context" (SURVEY 8(d)(ii)): pairs are independent, so
    # one pair per thread (ctypes releases the GIL during the call), every thread with its own arrays
    try:
        import threading
        orc.use_library(orc.build_native())
        n_thr = max(1, min(os.cpu_count() or 1, 256))
        GX, GY = orc.image_gradient(I1)
        work = [tuple(np.array(a) for a in (I0, D0, I1, GX, GY)) for _ in range(n_thr)]
        counts = [0] * n_thr
        budget = max(1.0, seconds * 0.2)
        start = threading.Barrier(n_thr + 1)

        def worker(k):
            i0, d0, i1, gx, gy = work[k]
            start.wait()
            t_end = time.perf_counter() + budget
            while time.perf_counter() < t_end:
                orc.dvo_normal_equations(i0, d0, i1, gx, gy, cam, cam, R, t, "huber")
                orc.photometric_error_sums(i0, d0, i1, cam, cam, T)
                counts[k] += 1
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_thr)]
        for th in threads:
            th.start()
        start.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t0
        out["c_port_O3_native_all_cores"] = dict(
            common, cores=n_thr, value=I0.size * sum(counts) / dt / 1e6, kind="port",
            sample=f"{n_thr} threads, one 640x480 pair each, {sum(counts)} DVO iterations in {dt:.1f} s; "
                   "oracle/tdk_oracle.c gcc -O3 -march=native, pair-parallel (the reference itself is single-threaded)")
    except Exception as e:                          # noqa: BLE001
        out["c_port_O3_native_all_cores"] = {"error": repr(e)}
    finally:
        orc.use_library(None)
    level = npp.Level(I0, D0, I1, cam, cam)
    n, dt = _timed_loop(lambda: npp.one_iteration(level, T, "huber"), seconds * 0.35)
    out["numpy_structured"] = dict(common, value=I0.size * n / dt / 1e6, kind="port",
                                   sample=f"1 pair {I0.shape[1]}x{I0.shape[0]}, {n} DVO iterations in {dt:.1f} s; "
                                          "oracle/numpy_port.py: the reference's own Python/NumPy structure (every "
                                          "intermediate materialised, lstsq on the masked M x 6 Jacobian)")
    return out


# ---------------------------------------------------------------------------
# the other BASELINE configs (one GPU)
# ---------------------------------------------------------------------------
def workload_dvo_single_pair(args, golden):
    """The drop-in path the examples drive: tadataka.vo.dvo.PoseChangeEstimator on host
    arrays, one 640x480 pair per call -- H2D, pyramid, 3 levels, D2H of the pose."""
    import tadataka_amd  # noqa: F401
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.vo import dvo
    from tadataka_amd import synthetic
    pair = synthetic.make_pair(480, 640, seed=0)
    cam = pair["cam"]
    cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
    est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
    weights = None if args.weights == "none" else args.weights
    if golden is not None and "plan_480x640_480x640_map" in golden:
        # the plans of the interpreter the fixture's reference run used (the default: this interpreter's own)
        def fixture_plans(shape, n_levels, ratio):
            from tadataka_amd import rescale_plan
            return rescale_plan.recorded_level_plans(golden, shape, n_levels, ratio)
        dvo.PYRAMID_PLANS = fixture_plans
    pose = est(pair["I0"], pair["D0"], pair["I1"], weights)          # creates the device batch
    n_calls = 100
    t0 = time.perf_counter()
    for _ in range(n_calls):
        pose = est(pair["I0"], pair["D0"], pair["I1"], weights)
    dt = time.perf_counter() - t0
    # kernel times in a second, profiled loop (profiling records events and waits once per launch)
    batch = dvo._batch_for((480, 640), 3, 1.5, False, dvo.PYRAMID)
    batch.set_profiling(True)
    for _ in range(20):
        est(pair["I0"], pair["D0"], pair["I1"], weights)
    prof = eval_profile(batch)
    batch.set_profiling(False)
    out = {"config": "BASELINE configs[1] through the drop-in API: tadataka.vo.dvo.PoseChangeEstimator, one "
                     "640x480 pair per call, host arrays in, Pose out (PCIe and launch latency included)",
           "ms_per_call": dt / n_calls * 1e3, "frame_pairs_per_s": n_calls / dt,
           "h2d_bytes_per_call": 3 * 480 * 640 * 8}
    tag = "v3_" + str(weights)
    if golden is not None and f"{tag}_t" in golden:
        from scipy.spatial.transform import Rotation
        err = max(float(np.max(np.abs(pose.R - Rotation.from_rotvec(golden[f"{tag}_rotvec"]).as_matrix()))),
                  float(np.max(np.abs(pose.t - golden[f"{tag}_t"]))))
        evals = golden[f"{tag}_evals"]
        shapes = [(213, 284), (320, 427), (480, 640)]
        px = sum(int(e) * h * w for e, (h, w) in zip(evals, shapes))
        out["pose_error_vs_reference_loop"] = err
        out["value"] = px * n_calls / dt / 1e6
        out["unit"] = "Mpx/s per DVO iter"
        assert err < 1e-6, f"single-pair pose differs from the reference loop by {err}"
    if prof["launches"]:
        kernel_ms = prof["total_ms"] / prof["launches"]
        out["roofline"] = roofline(BYTES_PER_PX_EVAL * prof["pixels"] / prof["launches"], kernel_ms,
                                   kernel="k_dvo_eval (full-resolution level, one pair: 300 blocks on 256 CUs)",
                                   bytes_per_px=BYTES_PER_PX_EVAL, launches=prof["launches"])
        out["roofline"]["note"] = ("latency, not bandwidth: one pair is 300 blocks of one round; what matters here is "
                                   "ms_per_call and its breakdown")
    # where a call's time goes (C ABI, each part synchronised on its own): upload of the three host arrays,
    # pyramid, estimation (the whole coarse-to-fine chain queued at once, one host wait)
    from tadataka_amd import _lib, ops

    def t_of(fn, n=100):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3
    ident = ops.pose12(np.eye(3), np.zeros(3))[None]
    mode = ops.WEIGHT_MODES[weights]
    out["breakdown_ms"] = {
        "upload_3_host_arrays": t_of(lambda: batch.upload(0, pair["I0"], pair["D0"], pair["I1"])),
        "pyramid": t_of(lambda: (batch.build_pyramid(), _lib.call("tdk_sync"))),
        "estimation": t_of(lambda: batch.estimate(cam, cam, ident, mode, 20)),
        "note": "7.4 MB over PCIe per call is the floor of the reference's signature (three fresh float64 arrays)"}
    return out


def workload_dvo_720p(args):
    """BASELINE configs[3], one GPU's shard: 64 pairs of 1280x720, 1 level, 2 iterations."""
    from tadataka_amd import ops, synthetic
    B, H, W = 64, 720, 1280
    cam = synthetic.camera_for(W, H)
    truth = true_poses(B, 0)
    batch = ops.DvoBatch(B, H, W, n_levels=1)
    batch.fill_synthetic(cam, truth, seed0=0, noise=0.02)
    ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    mode = ops.WEIGHT_MODES[None if args.weights == "none" else args.weights]
    for _ in range(2):
        batch.estimate(cam, cam, ident, mode, 2)
    batch.set_profiling(True)
    steps, px = 10, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        P, p = batch.estimate(cam, cam, ident, mode, 2)
        px += p
    dt = time.perf_counter() - t0
    prof = eval_profile(batch)
    batch.close()
    err0 = np.linalg.norm(truth[:, 9:], axis=1)
    err1 = np.linalg.norm(P[:, 9:] - truth[:, 9:], axis=1)
    kernel_ms = prof["total_ms"] / max(prof["launches"], 1)
    return {"config": "BASELINE configs[3], one GPU's shard: 64 pairs of 1280x720, 1 level, max_iter 2, "
                      f"weights={args.weights}, inputs generated on the device and resident: re-estimation at one level "
                      "(_PoseChangeEstimator), no rescale in the loop -- `--config cfg4` times the same shard with the "
                      "reference's rescale(., 1.0) of every new frame in the step",
            "value": px / dt / 1e6, "unit": "Mpx/s per DVO iter", "ms_per_step": dt / steps * 1e3,
            "frame_pairs_per_s": B * steps / dt,
            "median_translation_error_ratio": float(np.median(err1 / err0)),
            "roofline": roofline(BYTES_PER_PX_EVAL * prof["pixels"] / max(prof["launches"], 1), kernel_ms,
                                 kernel="k_dvo_eval (1280x720)", bytes_per_px=BYTES_PER_PX_EVAL,
                                 launches=prof["launches"])}


def workload_semi_dense(args, fixture):
    """BASELINE configs[2]: increment_age + propagate + update_depth on 640x480 maps with
    ~30 % valid pixels, B tracks per launch, everything resident in HBM."""
    from tadataka_amd import ops, synthetic
    B, H, W = 64, 480, 640
    N = H * W
    sd = ops.SemiDenseSession(B, H, W, max_refframes=2)
    sd.set_age_policy(False)       # the two halves run on the fixture's maps as they are (ages not limited)
    pg = ops.make_params(*SD_PARAMS)
    sd.set_params(pg, *SD_DEFAULTS)
    base = synthetic.make_semi_dense_case(H, W, seed=1)
    T10 = np.linalg.inv(base["T_wk"]) @ base["T_wr"]
    p_valid = 0.0
    for t in range(B):
        sd.push_frame(t, base["cam"], base["ref_image"], base["T_wr"])
        sd.push_frame(t, base["cam"], base["key_image"], base["T_wk"])
        if t == 0:
            age, pd_ = base["age"], base["prior_depth"]
        else:                      # same frames, another draw of the age map and of the prior noise
            rng = np.random.default_rng(1000 + t)
            age = (rng.uniform(0, 1, (H, W)) < 0.3).astype(np.uint64)
            pd_ = base["depth_gt"] * rng.uniform(0.9, 1.1, (H, W))
        p_valid += float((age > 0).mean()) / B
        sd.set_maps(t, pd_, base["prior_variance"], age)
    T10s = np.tile(T10, (B, 1, 1))
    sd.propagate(T10s, commit=False)
    hist = sd.update_depth(commit=False, histogram=True)
    if fixture is not None:
        assert np.array_equal(hist[0], fixture["flag_histogram"]), "track 0 does not reproduce the cfg3 fixture"
    reps, warp_ms, ud_ms = 10, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        sd.propagate(T10s, commit=False)
        warp_ms += sd.timing()["warp_ms"]
        sd.update_depth(commit=False)
        ud_ms += sd.timing()["update_depth_ms"]
    dt = time.perf_counter() - t0
    # the chained step of examples/semi_dense_vo.py:182-199 (update_depth on the PROPAGATED maps), not committed
    T_wk = np.tile(base["T_wk"], (B, 1, 1))
    sd.set_age_policy(True)        # one reference frame in the ring: ages saturate at 1
    sd.step(T10s, T_wk, commit=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        sd.step(T10s, T_wk, commit=False)
    dt_chain = time.perf_counter() - t0
    chain_ms = sd.timing()["step_ms"]
    fallbacks = sd.warp_fallbacks()
    sd.close()
    warp_ms /= reps
    ud_ms /= reps
    bytes_ud = 48.0 + 32.0 * p_valid
    out = {"config": "BASELINE configs[2]: increment_age + propagate + update_depth, 640x480, ~30 % valid pixels, "
                     f"{B} tracks per launch, device-resident session (tdk_sd)",
           "value": B * N * reps / dt / 1e6, "unit": "Mpx/s (map pixels through the three operators)",
           "frames_per_s": B * reps / dt, "ms_per_frame_host_api": dt / reps / B * 1e3,
           "timed_calls": "propagate (increment_age + propagate) and update_depth each on the SAME cfg3 input maps, "
                          "uncommitted -- the two halves of SURVEY 8(d) cfg3, which the fixture freezes; the chained "
                          "step (update_depth on the propagated maps) is `chained_step`",
           "chained_step": {"frames_per_s": B * reps / dt_chain, "ms_per_frame_host_api": dt_chain / reps / B * 1e3,
                            "kernel_ms_per_step": chain_ms},
           "valid_fraction": p_valid, "flag_histogram_track0": [int(v) for v in hist[0]],
           "roofline": roofline(bytes_ud * N * B, ud_ms, kernel="k_ud_classify + k_ud_estimate (update_depth)",
                                bytes_per_px=bytes_ud, tracks=B),
           "roofline_warp": roofline(BYTES_PER_PX_WARP * N * B, warp_ms,
                                     kernel="k_sd_targets + k_sd_gather2 (increment_age + propagate; slot path "
                                            "k_sd_scatter + k_sd_fold for tracks whose displacement box is too large: "
                                            "%d of this run)" % fallbacks,
                                     bytes_per_px=BYTES_PER_PX_WARP, tracks=B)}
    if not args.no_traffic_pass:
        got, note = measure_kernel_traffic([sys.executable, os.path.join(REPO, "tools", "sd_child.py"), "4", str(B)],
                                           {"warp": ("k_sd_targets", "k_sd_gather2"),
                                            "update_depth": ("k_ud_classify", "k_ud_estimate")})
        for key, label in (("roofline_warp", "warp"), ("roofline", "update_depth")):
            out[key]["traffic"] = got[label] if got else None
            out[key]["traffic_bytes_per_px"] = got[label] / (N * B) if got else None
            out[key]["traffic_source"] = note
    if not args.no_cpu_baseline:
        from oracle import oracle as orc             # the checker, here as the thing that is timed
        po = orc.make_params(*SD_PARAMS)
        key = (base["cam"], base["key_image"], base["T_wk"]); ref = (base["cam"], base["ref_image"], base["T_wr"])

        def one():
            orc.increment_age(base["age"], base["cam"], base["cam"], T10, base["prior_depth"])
            orc.propagate(T10, base["cam"], base["cam"], base["prior_depth"], base["prior_variance"], *SD_DEFAULTS)
            orc.update_depth(key, [ref], base["age"], base["prior_depth"], base["prior_variance"], po)
        n, cdt = _timed_loop(one, 2.0)
        out["cpu_baseline"] = {"value": N * n / cdt / 1e6, "unit": out["unit"], "cores": 1, "kind": "port",
                               "sample": f"{n} frames (the three operators on the track-0 maps) in {cdt:.1f} s, "
                                         "oracle/tdk_oracle.c -O2"}
    return out


def workload_dvo_stream(args):
    """A streaming consumer: the frame pairs do NOT sit in HBM for ever -- every step the new frame
    (I1) of each pair of the NEXT batch arrives over PCIe from pinned memory on that batch's copy
    stream while the CURRENT batch is estimated (tdk_dvo_upload_async[_u8]); I0 / D0 stay.  Two
    hand-over formats: float64 images as the reference's API carries them (2.46 MB per VGA frame), and
    8-bit grey frames as a camera delivers them, converted on the device (0.31 MB)."""
    from tadataka_amd import _lib, ops, synthetic
    B, H, W = args.pairs, 480, 640
    cam = synthetic.camera_for(W, H)
    mode = ops.WEIGHT_MODES[None if args.weights == "none" else args.weights]
    ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    out = {"config": f"BASELINE configs[1] as a stream: {B} pairs per batch, per step the I1 "
                     "frames of the next batch are uploaded from pinned host memory on a copy stream under the "
                     "current batch's estimation (three batches in flight: upload / pyramid / estimation); pyramid "
                     f"(--pyramid {args.pyramid}) + estimation as in the headline"}
    n_b = 3     # frames arrive for batch k + 2, the pyramid of batch k + 1 is built, batch k is estimated
    batches = []
    for k in range(n_b):
        bt = ops.DvoBatch(B, H, W, n_levels=3, ratio=1.5)
        if args.pyramid.startswith("skimage"):
            bt.set_skimage_pyramid(level0="all" if args.pyramid == "skimage" else ["D0"])
        else:
            bt.set_anti_aliasing(args.pyramid != "bilinear")
        bt.fill_synthetic(cam, true_poses(B, k * B), seed0=k * B, noise=0.02)
        bt.build_pyramid()                                        # (level 0 is a level of the pyramid in skimage mode)
        batches.append(bt)
    # "u8": only what changed is rebuilt (the levels of I1: tdk_dvo_build_pyramid_arrays); "u8_full_pyramid": every
    # array's levels per step as in rounds 2-3 (a consumer that also replaces I0 / D0 per step)
    for fmt, dtype, rebuild in (("f64", np.float64, ("I1",)), ("u8", np.uint8, ("I1",)), ("u8_full_pyramid", np.uint8, None)):
        pins = []
        for bt in batches:
            pin = ops.PinnedBuffer((B, H * W), dtype=dtype)
            i1 = np.stack([bt.download(i, 0, "I1").ravel() for i in range(0, B, max(B // 8, 1))])
            reps = (B + i1.shape[0] - 1) // i1.shape[0]
            src = np.tile(i1, (reps, 1))[:B]
            pin.array[:] = src if dtype == np.float64 else np.clip(np.rint(src * 255.0), 0, 255).astype(np.uint8)
            pins.append(pin)
        for bt in batches:
            bt.build_pyramid()                                    # I0 / D0 levels: once
        batches[0].upload_async("I1", 0, B, pins[0])
        batches[0].build_pyramid(rebuild)
        batches[1].upload_async("I1", 0, B, pins[1])

        def step(k):
            a, b, c = batches[k % n_b], batches[(k + 1) % n_b], batches[(k + 2) % n_b]
            c.upload_async("I1", 0, B, pins[(k + 2) % n_b])       # copy stream: PCIe only
            b.build_pyramid(rebuild)                              # its frames arrived during the last step
            a.estimate(cam, cam, ident, mode, args.max_iter)
        for k in range(n_b):
            step(k)
        _lib.call("tdk_sync")
        steps = 12 if dtype == np.uint8 else 6
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        _lib.call("tdk_sync")
        dt = time.perf_counter() - t0
        nbytes = B * H * W * np.dtype(dtype).itemsize
        out[fmt] = {"ms_per_step": dt / steps * 1e3, "frame_pairs_per_s": B * steps / dt,
                    "h2d_bytes_per_step": nbytes, "h2d_GBps_sustained": nbytes * steps / dt / 1e9,
                    "pyramid_arrays_rebuilt_per_step": list(rebuild) if rebuild else ["I0", "D0", "I1"]}
        for pin in pins:
            pin.close()
    for bt in batches:
        bt.close()
    out["note"] = ("f64: PCIe-bound (the float64 hand-over of the reference's API is 2.46 MB per frame); u8: the "
                   "upload hides under the estimation and only the new frame's pyramid levels are built, so the step "
                   "is shorter than the resident headline step (which rebuilds all three arrays of fresh pairs)")
    return out


def workload_semi_dense_dropin(args):
    """The mapping calls of examples/semi_dense_vo.py:182-199 exactly as the example writes them --
    Frame(...), increment_age, propagate, update_depth from rust_bindings.semi_dense, one 640x480 frame
    per iteration, the refframes list growing -- timed per frame on the host.  Eager ndarrays (the default: six
    maps downloaded per frame, as the reference returns them) and device maps (tadataka_amd.enable_device_maps():
    the maps stay on the device)."""
    import tadataka_amd  # noqa: F401
    import rust_bindings.semi_dense as rsd
    from rust_bindings.camera import CameraParameters
    from tadataka.matrix import inv_motion_matrix
    from tadataka_amd import _lib, synthetic
    H, W, n_frames = 480, 640, 9
    cam, depth0, T_w, images = synthetic.make_track(H, W, n_frames, step=(0.01, 0.002, 0.003))
    cp = CameraParameters((cam[0], cam[1]), (cam[2], cam[3]))
    params = rsd.Params(*SD_PARAMS)
    out = {"config": "examples/semi_dense_vo.py:182-199 through the unchanged rust_bindings.semi_dense calls "
                     "(Frame, increment_age, propagate, update_depth), 640x480, ages from zero as the example starts, one frame per "
                     "iteration, refframes growing; host wall time per frame, first frame (uploads of the caller's "
                     "initial maps) excluded"}
    for label, lazy in (("lazy_device_maps", True), ("eager_ndarrays", False)):
        rsd.LAZY_MAPS = lazy
        rng = np.random.default_rng(3)             # the same prior for both runs: their results must agree
        try:
            frame0 = rsd.Frame(cp, images[0], T_w[0])
            refframes = [frame0]
            depth_map0 = depth0 * rng.uniform(0.9, 1.1, (H, W))
            variance_map0 = np.full((H, W), 0.05)
            age0 = np.zeros((H, W), dtype=np.uint64)              # init_age of the example
            times = []
            for i in range(1, n_frames):
                transform10 = np.dot(inv_motion_matrix(T_w[i]), T_w[i - 1])
                _lib.call("tdk_sync")
                t0 = time.perf_counter()
                frame1 = rsd.Frame(cp, images[i], T_w[i])
                age1 = rsd.increment_age(age0, frame0.camera_params, frame1.camera_params, transform10, depth_map0)
                depth_map1, variance_map1 = rsd.propagate(transform10, frame0.camera_params, frame1.camera_params,
                                                          depth_map0, variance_map0, *SD_DEFAULTS)
                depth_map1, variance_map1, flag_map = rsd.update_depth(frame1, refframes, age1, depth_map1,
                                                                       variance_map1, params)
                refframes.append(frame1)
                depth_map0, variance_map0, age0 = depth_map1, variance_map1, age1
                frame0 = frame1
                _lib.call("tdk_sync")
                times.append(time.perf_counter() - t0)
            flags = np.asarray(flag_map)
            out[label] = {"ms_per_frame": float(np.median(times[1:])) * 1e3,
                          "ms_per_frame_all": [round(t * 1e3, 3) for t in times],
                          "success_pixels_last_frame": int((flags == 0).sum()),
                          "max_age": int(np.asarray(age0).max())}
            if lazy:
                # the same calls with nothing read back between frames (one wait at the end): the loop as a
                # caller that only consumes the final maps runs it -- no call waits for the device
                # (tdk_update_depth_maps knows from the host-side age bound that no age exceeds the refframes)
                frame0 = rsd.Frame(cp, images[0], T_w[0])
                refframes = [frame0]
                d0, v0, a0 = depth0 * 1.0, np.full((H, W), 0.05), np.zeros((H, W), dtype=np.uint64)
                T10s = [np.dot(inv_motion_matrix(T_w[i]), T_w[i - 1]) for i in range(1, n_frames)]
                t_start = None
                for rep in range(3):                      # ages keep growing with the refframes list
                    for i in range(1, n_frames):
                        if rep == 1 and i == 1:
                            _lib.call("tdk_sync")
                            t_start = time.perf_counter()
                        frame1 = rsd.Frame(cp, images[i], T_w[i])
                        a1 = rsd.increment_age(a0, frame0.camera_params, frame1.camera_params, T10s[i - 1], d0)
                        d1, v1 = rsd.propagate(T10s[i - 1], frame0.camera_params, frame1.camera_params, d0, v0, *SD_DEFAULTS)
                        d1, v1, f1 = rsd.update_depth(frame1, refframes, a1, d1, v1, params)
                        refframes.append(frame1)
                        d0, v0, a0, frame0 = d1, v1, a1, frame1
                _lib.call("tdk_sync")
                out[label]["ms_per_frame_pipelined"] = (time.perf_counter() - t_start) / (2 * (n_frames - 1)) * 1e3
        finally:
            rsd.LAZY_MAPS = False
    assert out["lazy_device_maps"]["success_pixels_last_frame"] == out["eager_ndarrays"]["success_pixels_last_frame"]
    return out


def workload_ba(args):
    """BASELINE configs[4]: 8 poses x 50 000 points, every point seen by every pose."""
    from tadataka_amd import ops, synthetic
    b = synthetic.make_ba_case()
    n = len(b["vp_idx"])
    x_obs = ops.ba_projection(b["poses"], b["points"], b["vp_idx"], b["pt_idx"], jacobians=False)
    ba = ops.BundleAdjustment(len(b["poses"]), len(b["points"]), b["vp_idx"], b["pt_idx"], x_obs)
    ba.block_sums(b["poses_noisy"], b["points_noisy"], per_point=False)
    ba.set_profiling(True)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ba.block_sums(b["poses_noisy"], b["points_noisy"], per_point=False)
    dt_sums = (time.perf_counter() - t0) / reps
    prof = ba.get_profile()
    red_ms = prof["block_reduce"][1] / max(prof["block_reduce"][0], 1)
    pts_ms = prof["point_sums"][1] / max(prof["point_sums"][0], 1)
    # the Levenberg-Marquardt loop from a start far enough away to need several iterations
    far = synthetic.make_ba_case(perturb=3e-2)
    ba.solve(far["poses_noisy"], far["points_noisy"], max_iter=20)      # warm-up
    # what a call costs before the first iteration: parameters up (1.2 MB), block sums, parameters down
    ba.set_profiling(False)
    dt_fixed = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ba.solve(far["poses_noisy"], far["points_noisy"], max_iter=0)
        dt_fixed = min(dt_fixed, time.perf_counter() - t0)
    # thresholds 0: all 6 iterations run (the last ones at the rounding floor of the error)
    ba.set_profiling(False)
    lm_kw = dict(max_iter=6, absolute_error_threshold=0.0, relative_error_threshold=0.0)
    dt_solve = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        poses, points, errors = ba.solve(far["poses_noisy"], far["points_noisy"], **lm_kw)
        dt_solve = min(dt_solve, time.perf_counter() - t0)
    ba.set_profiling(True)                               # per-kernel times from a separate, event-instrumented call
    ba.solve(far["poses_noisy"], far["points_noisy"], **lm_kw)
    lm = ba.get_profile()
    ba.close()
    iters = max(len(errors) - 1, 1)
    trials = max(lm["schur"][0], 1)
    out = {"config": "BASELINE configs[4]: local BA window, 8 poses x 50 000 points, 400 000 observations",
           "value": n / (dt_sums) / 1e6, "unit": "Mobs/s (block sums U, ea | V, eb per call: parameters uploaded, per-pose sums downloaded, per-point sums left in HBM)",
           "block_sums_ms_host_api": dt_sums * 1e3,
           "lm_iterations": iters, "lm_damping_trials": trials, "lm_call_ms": dt_solve * 1e3,
           "lm_call_fixed_ms": dt_fixed * 1e3,
           "lm_ms_per_iteration": (dt_solve - dt_fixed) / iters * 1e3,
           "lm_ms_per_damping_trial": (dt_solve - dt_fixed) / trials * 1e3,
           "lm_initial_mean_squared_error": float(errors[0]), "lm_final_mean_squared_error": float(errors[-1]),
           "lm_kernel_ms": {k: (v[1] / v[0] if v[0] else 0.0) for k, v in lm.items()},
           "roofline": roofline(BYTES_PER_OBS_BA * n, red_ms + pts_ms,
                                kernel="k_ba_reduce_seg<STORE_B> + k_ba_point_sums (U, ea | V, eb; no atomics)",
                                bytes_per_obs=BYTES_PER_OBS_BA, block_reduce_ms=red_ms, point_sums_ms=pts_ms)}
    # 22 MB of observations per launch: this workload is a chain of latency-bound kernels, not a bandwidth problem --
    # the budget that matters is microseconds per kernel against the ~1.5-1.9 us a dependent kernel boundary costs
    # (MI355X_MICROARCH.md) plus one chain of L2 misses (~1 us) per dependent access level
    kern_us = {k: v * 1e3 for k, v in out["lm_kernel_ms"].items()}
    out["latency_budget"] = {
        "per_kernel_us": kern_us, "kernels_per_damping_trial": len([v for v in kern_us.values() if v > 0]),
        "sum_of_kernels_us_per_trial": sum(kern_us.values()),
        "lm_us_per_damping_trial": out["lm_ms_per_damping_trial"] * 1e3,
        "kernel_boundary_us": 1.7,
        "note": "the roofline fraction of this workload says only that 22 MB do not fill the machine: every kernel of a "
                "damping trial takes 4 - 35 us of dependent-latency (one block per pose segment / one workgroup for the "
                "48 x 48 reduced camera system), launches are ~3 % of the iteration"}
    if not args.no_cpu_baseline:
        from oracle import oracle as orc             # the checker, here as the thing that is timed

        def one():
            orc.ba_block_reduce(b["poses_noisy"], b["points_noisy"], x_obs, b["vp_idx"], b["pt_idx"])
        k, cdt = _timed_loop(one, 2.0)
        out["cpu_baseline"] = {"value": n * k / cdt / 1e6, "unit": "Mobs/s", "cores": 1, "kind": "port",
                               "sample": f"{k} block reduces of the same graph in {cdt:.1f} s, oracle/tdk_oracle.c -O2"}
    return out


# ---------------------------------------------------------------------------
def spawn_ranks(n, dry=False):
    """`python bench.py --gpus N` by itself: one worker process per GPU with the
    environment torch.distributed.run would export; rank 0's JSON line is relayed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]                   # only a rendezvous key: nothing listens on it
    import tempfile
    procs, logs = [], []
    logdir = tempfile.mkdtemp(prefix="tdk_bench_ranks_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if dry:
            env["TDK_BENCH_DRY"] = "1"
        # every rank's stdout / stderr is kept (rank 0's stdout is the JSON line) so that a failed
        # rank can be named with what it last said instead of an exit code only
        o = open(os.path.join(logdir, "rank%d.out" % r), "w+")
        e = open(os.path.join(logdir, "rank%d.err" % r), "w+")
        logs.append((o, e))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=o, stderr=e, text=True))
    rcs = [p.wait() for p in procs]

    def tail(f, k):
        f.flush()
        f.seek(0)
        return f.read().splitlines()[-k:]

    sys.stdout.write("".join(line + "\n" for line in tail(logs[0][0], 10 ** 9)))
    sys.stdout.flush()
    failed = any(rcs)
    for r, (o, e) in enumerate(logs):
        lines = tail(e, 10 ** 9 if (r == 0 and not failed) else 15)
        if failed:
            sys.stderr.write("---- rank %d: exit code %d, LOCAL_RANK=%d (HIP device LOCAL_RANK %% device count), "
                             "last stderr lines:\n" % (r, rcs[r], r))
            if r > 0:
                lines += ["(stdout) " + s for s in tail(o, 5)]
        for s in lines:
            sys.stderr.write(("    " if failed else "") + s + "\n")
        o.close()
        e.close()
    sys.stderr.flush()
    if failed:
        raise SystemExit("bench worker exit codes: %s (per-rank logs: %s)" % (rcs, logdir))
    import shutil
    shutil.rmtree(logdir, ignore_errors=True)


def main():
    args = parse_args()
    if args.dry_ranks > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.dry_ranks, dry=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("NCCL_DEBUG", "WARN")     # RCCL says on stderr why an init failed

    from tadataka_amd import _lib, ops, rescale_plan, sharding, synthetic
    _lib.require_gpu()
    # one GPU per rank; on a box with fewer GPUs than ranks (only ever a smoke test) ranks share devices
    _lib.call("tdk_set_device", local_rank % _lib.device_count())       # (= `device` below)
    # RCCL through the C ABI when WORLD_SIZE > 1.  Only where ranks have to SHARE a GPU (fewer GPUs than
    # ranks: the one-GPU smoke test of the multi-process path, RCCL refuses duplicate devices) may the few
    # bytes of poses and scalars go through files instead -- the JSON line says so ("exchange").  With a GPU
    # per rank a transport failure is an error: a scaling run must not "pass" without RCCL.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    device = local_rank % _lib.device_count()
    if os.environ.get("TDK_BENCH_DRY") == "1":
        comm = sharding.FileComm(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
        comm_error = "dry run (--dry-ranks): the ranks share the GPU, RCCL was not attempted"
    else:
        try:
            comm, comm_error = sharding.connect_or_fallback(allow_file_fallback=_lib.device_count() < local_world)
        except sharding.TransportUnavailable as e:
            # a scaling run must not pass without RCCL: say where this rank stood, and fail
            sys.stderr.write(
                "bench.py rank %s (local rank %d): RCCL could not be brought up: %s\n"
                "  device bound: %d of %d visible (%s); HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s "
                "CUDA_VISIBLE_DEVICES=%s HSA_ENABLE_IPC_MODE_LEGACY=%s NCCL_DEBUG=%s\n"
                "  (RCCL's own warnings are above this line on stderr)\n"
                % (os.environ.get("RANK", "0"), local_rank, e, device, _lib.device_count(), _lib.device_name(),
                   os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"),
                   os.environ.get("CUDA_VISIBLE_DEVICES"), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                   os.environ.get("NCCL_DEBUG")))
            sys.stderr.flush()
            raise SystemExit(3)
    if comm_error:
        sys.stderr.write("bench.py: RCCL unavailable (%s); exchanging poses through files\n" % comm_error)
    world, rank = comm.world, comm.rank

    B, H, W = args.pairs, args.height, args.width
    cam = synthetic.camera_for(W, H)
    weights = None if args.weights == "none" else args.weights
    # Everything that is not the headline runs FIRST -- the CPU baselines (rank 0, one process), then the other
    # BASELINE configs -- so that the tail of the run is the timed GPU work of the headline.
    early = {}
    golden_early = None
    skimage_mode = args.pyramid.startswith("skimage")
    gpath = os.path.join(REPO, "tests", "golden", "skimage_dvo.npz" if skimage_mode else "dvo_vga_pyramid.npz")
    is_cfg2 = (H, W, args.levels, args.max_iter) == (480, 640, 3, 20)
    if is_cfg2 and os.path.exists(gpath):
        golden_early = np.load(gpath)
    # the plans of the pyramid: the fixture's for configs[1] (every rank: pair 0's assertion is against the reference
    # run on that interpreter's scikit-image), this interpreter's own otherwise
    fixture_plans = None
    if skimage_mode and golden_early is not None:
        fixture_plans = rescale_plan.recorded_level_plans(golden_early, (H, W), args.levels, 1.5)
    if rank != 0:
        golden_early = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        hp = synthetic.make_pair(H, W, seed=0)
        early["cpu_baselines"] = dvo_cpu_baselines(hp["I0"], hp["D0"], hp["I1"], cam, args.cpu_seconds)
    if rank == 0 and world == 1 and not args.no_traffic_pass and os.environ.get("TDK_BENCH_DRY") != "1":
        early["traffic"] = measure_traffic_in_run(sys.argv[1:])
    if rank == 0 and world == 1 and not args.no_workloads:
        fixture = None
        fpath = os.path.join(REPO, "tests", "golden", "semi_dense_cfg3.npz")
        if os.path.exists(fpath):
            fixture = np.load(fpath)
        wl = {}
        for name, fn in (("dvo_single_pair_vga", lambda: workload_dvo_single_pair(args, golden_early)),
                         ("dvo_720p_x64", lambda: workload_dvo_720p(args)),
                         ("dvo_stream_x256", lambda: workload_dvo_stream(args)),
                         ("semi_dense_vga", lambda: workload_semi_dense(args, fixture)),
                         ("semi_dense_dropin_vga", lambda: workload_semi_dense_dropin(args)),
                         ("ba_8x50k", lambda: workload_ba(args))):
            try:
                wl[name] = fn()
            except AssertionError:
                raise
            except Exception as e:              # noqa: BLE001
                wl[name] = {"error": repr(e)}
        early["workloads"] = wl
    mode = ops.WEIGHT_MODES[weights]
    anti_aliasing = args.pyramid != "bilinear"
    n_batches = 2 if args.double_buffer else 1
    # this rank's shard of the pair ids: n_batches consecutive blocks of B pairs
    seeds = [sharding.batch_seed0(rank, n_batches, B, k) for k in range(n_batches)]
    batches = []
    for seed0 in seeds:
        bt = ops.DvoBatch(B, H, W, n_levels=args.levels, ratio=1.5)
        if skimage_mode:
            bt.set_skimage_pyramid(fixture_plans, level0="all" if args.pyramid == "skimage" else ["D0"])
        else:
            bt.set_anti_aliasing(anti_aliasing)
        bt.fill_synthetic(cam, true_poses(B, seed0), seed0=seed0, noise=0.02)
        batches.append(bt)
    batch = batches[0]
    # pair 0 of the whole job = the pair the reference's own PoseChangeEstimator was run on
    golden, host_pair = golden_early, None
    if golden is not None:
        host_pair = synthetic.make_pair(H, W, seed=0)
        batch.upload(0, host_pair["I0"], host_pair["D0"], host_pair["I1"])
    ident = np.tile(ops.pose12(np.eye(3), np.zeros(3)), (B, 1))
    counter = [0]
    work_px = [0, 0]          # source pixels of PhotometricError evaluations / of pose updates (device counters)
    pair0_pose = [None]
    gather = sharding.PoseGather(B, comm)
    gather_host_s = [0.0]     # host time in finish() of the previous step's gather + start() of this one's
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
        e_px, u_px = cur.counts()
        work_px[0] += e_px
        work_px[1] += u_px
        if k % n_batches == 0:
            pair0_pose[0] = poses[0].copy()             # the golden pair lives in batch 0
        # the only exchange: the recovered poses, all-gathered (RCCL over xGMI) from where the
        # device loop left them.  The gather of this step is queued now and collected after the
        # next step's estimation (flush() collects the last one inside the timed region).
        t_g = time.perf_counter()
        previous = gather.finish() if gather.pending else None
        gather.start(poses, cur)
        gather_host_s[0] += time.perf_counter() - t_g
        return previous, px, k % n_batches

    def fence():
        _lib.call("tdk_sync")
        comm.barrier()
        _lib.call("tdk_sync")

    for _ in range(args.warmup):
        step()
    if gather.pending:
        gather.finish()
    for bt in batches:
        bt.set_profiling(True)
    work_px[0] = work_px[1] = 0
    gather_host_s[0] = 0.0

    def timed_block():
        fence()
        t0 = time.perf_counter()
        pixels, last = 0, 0
        for _ in range(args.steps):
            _, px, last = step()
            pixels += px
        poses = gather.finish() if gather.pending else None   # all-gathered poses of the last step
        fence()
        dt = time.perf_counter() - t0
        dt = float(sharding.reduce_scalars([dt], "max", comm)[0])
        return dt, pixels, poses, last

    elapsed, pixels, poses, last_batch = timed_block()
    blocks = 1
    n_more = max(0, int(math.ceil(args.min_seconds / max(elapsed, 1e-6))) - 1)
    n_more = int(sharding.reduce_scalars([float(n_more)], "max", comm)[0])     # every rank repeats alike
    for _ in range(n_more):
        dt, px, poses, last_batch = timed_block()
        elapsed += dt
        pixels += px
        blocks += 1
    total_steps = blocks * args.steps
    # full-resolution k_dvo_eval launches by what they evaluated: in full, probes, both
    prof_kind = {k: {"launches": 0, "total_ms": 0.0, "pixels": 0} for k in ("full", "probe", "mixed")}
    for bt in batches:
        for kind in prof_kind:
            for key, val in bt.get_profile(kind).items():
                prof_kind[kind][key] += val
        bt.set_profiling(False)
    prof = {key: sum(prof_kind[k][key] for k in prof_kind) for key in ("launches", "total_ms", "pixels")}
    # A clean per-level picture, outside the timed region: ONE batch alone on the device (no other batch's
    # pyramid beside it), every level's evaluation launches between HIP events, and the pyramid build timed
    # on its own.  With two batches in flight the coarse-level launches are stretched by the other batch's
    # pyramid kernel; these numbers are what each kernel takes by itself.
    by_level, pyramid_alone = {}, None
    if rank == 0 and not args.no_solo_pass:
        _lib.call("tdk_sync")                       # (no collective here: only rank 0 takes this pass)
        solo = batches[0]
        solo.set_profiling(True, all_levels=True)
        for _ in range(3):
            solo.build_pyramid()
            solo.estimate(cam, cam, ident, mode, args.max_iter)
        for lv in range(args.levels):
            entry = {}
            for kind in ("full", "probe", "mixed"):
                pk = solo.get_profile(kind, level=lv)
                if pk["launches"]:
                    kms = pk["total_ms"] / pk["launches"]
                    r = roofline(BYTES_PER_PX_EVAL * pk["pixels"] / pk["launches"], kms, launches=pk["launches"])
                    entry[kind] = {k: r[k] for k in ("achieved", "frac", "kernel_ms", "launches")}
                    entry[kind]["px_per_launch"] = pk["pixels"] / pk["launches"]
            h_l, w_l = solo.level_shape(lv)
            by_level[f"level{lv}_{w_l}x{h_l}"] = entry
        solo.set_profiling(False)
        if args.levels > 1 or solo.level0_mask:
            _lib.call("tdk_sync")
            n_builds = 20
            t0 = time.perf_counter()
            for _ in range(n_builds):
                solo.build_pyramid()
            _lib.call("tdk_sync")
            pms = (time.perf_counter() - t0) / n_builds * 1e3
            out_px = sum(solo.level_shape(lv)[0] * solo.level_shape(lv)[1] for lv in range(1, args.levels))
            n_l0 = bin(solo.level0_mask & 7).count("1")       # arrays whose level 0 is a rescale of its own: + one write each
            pbytes = 8.0 * B * (3 * (H * W + out_px) + n_l0 * H * W)   # I0, D0, I1: frame read once, the levels written
            pyramid_alone = roofline(pbytes, pms, kernel="pyramid build of one batch alone (k_pyramid_stream / "
                                     "k_rescale_aa_multi for anti-aliased levels), host-timed over %d builds" % n_builds,
                                     bytes_note="compulsory traffic: 3 arrays x (frame read once + levels written), "
                                                "+ one frame written per array whose level 0 is a rescale of its own")
    # The same step with the other pyramid readings, a few steps each, AFTER the headline (every rank: step() holds
    # the collective): what the reference's level 0 + clip cost, and the reading rounds 1-4 measured.
    alt_pyramids = {}
    pair0_headline = None if pair0_pose[0] is None else pair0_pose[0].copy()
    if skimage_mode and not args.no_solo_pass:
        for label in ("skimage-depth-level0", "ideal"):
            for bt in batches:
                bt.set_profiling(False)
                if label == "ideal":
                    bt.set_ideal_pyramid()
                else:
                    bt.set_skimage_pyramid(fixture_plans, level0=["D0"])
            for _ in range(max(2, args.warmup)):
                step()
            if gather.pending:
                gather.finish()
            n_alt = max(4, min(args.steps, 20))
            fence()
            t0 = time.perf_counter()
            for _ in range(n_alt):
                step()
            if gather.pending:
                gather.finish()
            fence()
            dt_alt = float(sharding.reduce_scalars([time.perf_counter() - t0], "max", comm)[0])
            alt_pyramids[label] = {"ms_per_step": dt_alt / n_alt * 1e3, "steps": n_alt}
    # which device every rank ran on (a one-hot sum: rank r contributes its HIP device index at position r)
    dev = C.c_int()
    _lib.call("tdk_get_device", C.byref(dev))
    onehot = np.zeros(world)
    onehot[rank] = float(dev.value)
    rank_devices = [int(v) for v in sharding.reduce_scalars(onehot, "sum", comm)]
    gather_us_max = float(sharding.reduce_scalars([gather_host_s[0] / max(total_steps, 1) * 1e6], "max", comm)[0])
    pixels_all, error_px_all, update_px_all = (float(v) for v in sharding.reduce_scalars(
        [float(pixels), float(work_px[0]), float(work_px[1])], "sum", comm))

    if rank == 0:
        # the all-gathered poses of the last step: batch `last_batch` of every rank, in rank order
        truth = np.concatenate([true_poses(B, s0) for s0 in sharding.gathered_seed0s(world, n_batches, B, last_batch)])
        assert poses.shape == truth.shape
        t_err = float(np.max(np.linalg.norm(poses[:, 9:] - truth[:, 9:], axis=1)))
        kernel_ms = prof["total_ms"] / max(prof["launches"], 1)
        pmc = load_profile_json("pmc_dvo_eval.json") or {}
        traffic = pmc.get("hbm_bytes_per_launch")
        measured, measured_note = early.get("traffic", (None, "not attempted (--no-traffic-pass, or more than one rank)"))
        # the dominant kernel: k_dvo_eval's full evaluations at full resolution (by_mode.full); the launch-weighted
        # mix with the error-only probes -- what rounds 2-5 reported as `frac` -- stays as `mix`
        dom = prof_kind["full"] if prof_kind["full"]["launches"] else prof
        dom_ms = dom["total_ms"] / max(dom["launches"], 1)
        rl = roofline(BYTES_PER_PX_EVAL * dom["pixels"] / max(dom["launches"], 1), dom_ms,
                      kernel=f"k_dvo_eval<{args.weights}>, full evaluations at full resolution (by_mode.full); `mix`: every "
                             "full-resolution launch incl. the error-only k_dvo_probe",
                      bytes_per_px=BYTES_PER_PX_EVAL,
                      px_per_launch=dom["pixels"] / max(dom["launches"], 1), launches=dom["launches"],
                      limiter="FP64 issue at the package power cap (DESIGN.md 5.1); the probes (error only) are HBM-bound")
        mix = roofline(BYTES_PER_PX_EVAL * prof["pixels"] / max(prof["launches"], 1), kernel_ms)
        rl["mix"] = {k: mix[k] for k in ("achieved", "frac", "kernel_ms")}
        rl["mix"]["launches"] = prof["launches"]
        if measured:
            rl["traffic"] = measured["hbm_bytes_per_launch"]
            rl["traffic_source"] = "measured in this run: " + measured_note
            rl["traffic_detail"] = dict(measured, committed_profile_bytes_per_launch=traffic)
        else:
            rl["traffic"] = traffic
            rl["traffic_source"] = ("profiles/pmc_dvo_eval.json (%s): separate rocprofv3 --pmc passes over this command, "
                                    "not measured in this run (%s)" % (pmc.get("source", "?"), measured_note)) if traffic else None
        rl["timing_note"] = ("kernel_ms: HIP events recorded on the batch's own stream around every full-resolution "
                             "launch INSIDE the timed region (one event pair + one host wait per launch: the headline "
                             "is measured with that overhead, i.e. conservatively)")
        by_mode = {}
        for kind, pk in prof_kind.items():
            if pk["launches"]:
                kms = pk["total_ms"] / pk["launches"]
                r = roofline(BYTES_PER_PX_EVAL * pk["pixels"] / pk["launches"], kms, launches=pk["launches"])
                by_mode[kind] = {k: r[k] for k in ("achieved", "frac", "kernel_ms", "launches")}
        rl["by_mode"] = by_mode
        rl["by_level"] = by_level
        rl["by_level_note"] = ("one batch alone on the device after the timed region (single-buffer, 3 steps, HIP events "
                               "around every level's launches); by_mode / kernel_ms above are from inside the timed region")
        out = {
            "metric": "warp+residual+JtJ Mpixels/sec per DVO iter",
            "value": update_px_all / elapsed / 1e6,
            "unit": "Mpx/s",
            "value_definition": "SURVEY 8(d): one DVO iter = one calc_pose_update + one photometric_error over a "
                                "level's source pixels; value = level pixels x pose updates solved / wall time "
                                "(the one extra error evaluation per level and the pyramid are in the time, not in "
                                "the count)",
            "update_mpx_per_s": update_px_all / elapsed / 1e6,
            "error_mpx_per_s": error_px_all / elapsed / 1e6,
            "error_evaluations_mpx_per_s": pixels_all / elapsed / 1e6,
            "updates_per_step": update_px_all / total_steps / world / (B * H * W),
            "errors_per_step": error_px_all / total_steps / world / (B * H * W),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / total_steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"name": args.config + (" (BASELINE configs[3]: one GPU's shard of the 512 x 1280x720 batch)"
                                              if args.config == "cfg4" else " (BASELINE configs[1])"),
                       "workload": "DVO pose estimation (PoseChangeEstimator), batch of independent "
                                   f"{W}x{H} frame pairs, {args.levels}-level pyramid ratio 1.5, "
                                   f"weights={args.weights}, max_iter={args.max_iter}",
                       "pairs_per_gpu": B, "batches_in_flight": n_batches,
                       "pyramid": {"skimage": "skimage.transform.rescale to the bit, every level incl. level 0 (rescale(., 1.0)), "
                                              "clip=True; plans (estimated affine maps, scipy kernels): " +
                                              ("tests/golden/skimage_dvo.npz (scikit-image 0.18.3 / numpy 1.26.4)"
                                               if fixture_plans else "this interpreter's"),
                                   "skimage-depth-level0": "as skimage, level 0 of its own for the depth map only",
                                   "ideal": "anti-aliased at the ideal sample positions, level 0 = the frame, no clip",
                                   "bilinear": "bilinear at the ideal sample positions"}[args.pyramid],
                       "height": H, "width": W, "levels": args.levels,
                       "weights": args.weights, "max_iter": args.max_iter,
                       "parallelism": f"pair-shard x{world}, RCCL all-gather of poses" if world > 1 else "single GPU"},
            "timed_blocks": blocks, "timed_seconds": elapsed,
            "frame_pairs_per_s": B * world * total_steps / elapsed,
            "dvo_iterations_per_pair_per_step": pixels / total_steps / B / (H * W),
            "max_translation_error": t_err,
            "rccl_ranks": world if (world > 1 and comm.kind == "rccl") else 0,
            "rank_devices": rank_devices,     # HIP device index of rank 0, 1, ... (one process per GPU: all different)
            "device_name": _lib.device_name(),
            "pose_gather_host_us_per_step": gather_us_max,   # finish(step k - 1) + start(step k), max over ranks
            "exchange": {"rccl": "ncclAllGather of the device-resident poses (tdk_comm, C ABI)",
                         "file": "files in TMPDIR -- %s" % comm_error,
                         "local": "none (one process)"}[comm.kind],
            "roofline": rl,
        }
        rf = roofline_fp64(prof_kind)
        if rf:
            out["roofline_fp64"] = rf
        if pyramid_alone:
            out["pyramid_roofline"] = pyramid_alone
        if alt_pyramids:
            alt_pyramids["note"] = ("the same step, same batches, after the timed region: 'skimage-depth-level0' = level 0 of "
                                    "its own for the depth map only (poses identical to 1e-16, DESIGN.md 3); 'ideal' = ideal "
                                    "sample positions, level 0 = the frame, no clip (what rounds 1-4 measured" +
                                    (": BENCH_r04 2.73 ms)" if is_cfg2 else ")"))
            out["other_pyramid_readings"] = alt_pyramids

        if golden is not None and pair0_headline is not None:
            from scipy.spatial.transform import Rotation
            tag = ("v3_" if skimage_mode else ("pyr_aa_" if anti_aliasing else "pyr_")) + str(weights)
            if f"{tag}_t" in golden:
                p0 = pair0_headline                     # pair 0 of batch 0, from the last headline step that ran it
                err = max(float(np.max(np.abs(p0[:9].reshape(3, 3) -
                                              Rotation.from_rotvec(golden[f"{tag}_rotvec"]).as_matrix()))),
                          float(np.max(np.abs(p0[9:] - golden[f"{tag}_t"]))))
                out["pair0_pose_error_vs_reference_loop"] = err
                out["pair0_reference"] = ("the reference's own PoseChangeEstimator on the REAL skimage.transform.rescale "
                                          "(scikit-image 0.18.3; tests/golden/skimage_dvo.npz)" if skimage_mode else
                                          "the reference's own PoseChangeEstimator on the ideal-constants stand-in rescale "
                                          "(tests/golden/dvo_vga_pyramid.npz)")
                assert err < 1e-6, f"pair 0 differs from the reference's PoseChangeEstimator by {err}"
        if "cpu_baselines" in early:
            cb = early["cpu_baselines"]
            out["cpu_baselines"] = cb
            best = cb.get("c_port_O3_native")
            if not best or "value" not in best:
                best = cb["c_port_O2_exact"]
            out["cpu_baseline"] = best
            out["speedup_vs_cpu_baseline"] = out["value"] / best["value"]
            out["speedup_vs_numpy_structured_reference_path"] = out["value"] / cb["numpy_structured"]["value"]
        for bt in batches:
            bt.close()
        batches = []
        if "workloads" in early:
            out["workloads"] = early["workloads"]
        out["run_order"] = "cpu baselines, HBM-traffic passes (child runs under rocprofv3 --pmc), other workloads, headline (timed region last)"
        print(json.dumps(out))
    for bt in batches:
        bt.close()
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
