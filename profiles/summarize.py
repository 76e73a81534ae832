#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace + PMC passes written by
profiles/run_profile.sh) into a text summary and profiles-ready JSON.

Usage: summarize.py <dir written by run_profile.sh> [pixels_per_level0_launch]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KiB units); on gfx950
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, i.e. half
of the bytes, so the read side is doubled.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def kernel_stats(root, sub="trace"):
    rows = []
    for f in find(os.path.join(root, sub), "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    return rows


def trace_by_grid(root, sub="trace"):
    """(kernel, grid) -> [durations ns]"""
    acc = defaultdict(list)
    for f in find(os.path.join(root, sub), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            acc[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc


def pmc_by_grid(root, sub):
    """(kernel, total grid size) -> counter -> [values]"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(os.path.join(root, sub), "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    root = sys.argv[1]
    px = float(sys.argv[2]) if len(sys.argv) > 2 else 256 * 480 * 640
    out = {"pixels_per_level0_launch": px}
    is_eval = lambda name: name.startswith("k_dvo_eval") or name.startswith("k_dvo_probe")
    levels = []
    for sub, label in (("trace", "UNDER OVERLAP: two batches in flight (the headline's mode); only the full-resolution "
                                 "evaluation rows are kernel times, the others run beside the other batch's pyramid"),
                       ("trace_single", "ALONE: --single-buffer, one batch, every kernel by itself on the device")):
        rows = kernel_stats(root, sub)
        if not rows:
            continue
        print(f"== rocprofv3 --kernel-trace --stats -- {label} ==")
        print(f'{"kernel":42s} {"calls":>6s} {"avg_us":>10s} {"total_ms":>10s} {"pct":>7s}')
        stats = []
        for r in rows:
            stats.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                          "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                          "pct": float(r["Percentage"])})
            s = stats[-1]
            print(f'{s["kernel"][:42]:42s} {s["calls"]:6d} {s["avg_us"]:10.1f} {s["total_ms"]:10.2f} {s["pct"]:7.2f}')
        out["kernel_stats" if sub == "trace" else "kernel_stats_alone"] = stats
        print(f"\n-- k_dvo_eval / k_dvo_probe by grid (one grid per pyramid level; largest = full resolution), "
              f"{'under overlap' if sub == 'trace' else 'alone'}; HBM roofline on 24 B/px --")
        tr = trace_by_grid(root, sub)
        evals = sorted([(k, v) for k, v in tr.items() if is_eval(k[0])],
                       key=lambda kv: (-kv[0][1][0] * kv[0][1][1], kv[0][0]))
        lv = []
        # pixels per launch of a level follow from the grid ratio to the finest level's (same pairs, px ~ blocks)
        top_grid = max((g[0] * g[1] for (_, g), _ in evals), default=0)
        for (name, grid), durs in evals:
            rec = {"kernel": name, "grid": list(grid), "launches": len(durs), "avg_us": mean(durs) / 1e3,
                   "min_us": min(durs) / 1e3, "max_us": max(durs) / 1e3}
            lv.append(rec)
            print(f'{name:28s} grid={grid} launches={len(durs)} avg={mean(durs)/1e3:9.1f} us '
                  f'min={min(durs)/1e3:9.1f} max={max(durs)/1e3:9.1f}')
        out["dvo_eval_levels" if sub == "trace" else "dvo_eval_levels_alone"] = lv
        if sub == "trace":
            levels = lv
        print()

    fetch = pmc_by_grid(root, "pmc_fetch")
    write = pmc_by_grid(root, "pmc_write")
    # every full-resolution evaluation launch: the combined kernel (full evaluations, mixed rounds) and the probe
    # kernel -- what bench.py's `roofline` averages over; per kernel below
    ev_f = sorted([(k, v) for k, v in fetch.items() if is_eval(k[0])], key=lambda kv: -kv[0][1])
    ev_w = sorted([(k, v) for k, v in write.items() if is_eval(k[0])], key=lambda kv: -kv[0][1])
    if ev_f and ev_w and levels:
        top = ev_f[0][0][1]
        fall = [x for k, v in ev_f if k[1] == top for x in v["FETCH_SIZE"]]
        wall = [x for k, v in ev_w if k[1] == top for x in v["WRITE_SIZE"]]
        fs = mean(fall)       # KiB, raw
        ws = mean(wall)
        hbm = (2.0 * fs + ws) * 1024.0
        top_levels = [l for l in levels if l["grid"] == levels[0]["grid"]]
        avg_s = sum(l["avg_us"] * l["launches"] for l in top_levels) / sum(l["launches"] for l in top_levels) * 1e-6
        for (name, grid), v in ev_f:
            if grid == top:
                w_ = [x for (n2, g2), v2 in ev_w if n2 == name and g2 == grid for x in v2["WRITE_SIZE"]]
                b_ = (2.0 * mean(v["FETCH_SIZE"]) + (mean(w_) if w_ else 0.0)) * 1024.0
                print(f"   {name:28s} {b_/1e9:.3f} GB per launch = {b_/px:.2f} B/px")
                out.setdefault("pmc_level0_by_kernel", {})[name] = {"hbm_bytes_per_launch": b_, "hbm_bytes_per_px": b_ / px}
        out["pmc_level0"] = {"FETCH_SIZE_KiB_raw": fs, "WRITE_SIZE_KiB_raw": ws,
                             "hbm_bytes_per_launch": hbm, "hbm_bytes_per_px": hbm / px,
                             "algorithmic_bytes_per_px": 24.0,
                             "note": "read side doubled per the gfx950 FETCH_SIZE correction"}
        print(f"\n== HBM traffic of the full-resolution evaluation launches (k_dvo_eval + k_dvo_probe, launch-weighted) ==")
        print(f"FETCH_SIZE raw {fs:.0f} KiB  WRITE_SIZE raw {ws:.0f} KiB  -> {hbm/1e9:.3f} GB per launch "
              f"= {hbm/px:.2f} B/px (algorithmic 24 B/px); at {avg_s*1e6:.0f} us: "
              f"{hbm/avg_s/1e9:.0f} GB/s moved, {24.0*px/avg_s/1e9:.0f} GB/s algorithmic")
        json.dump({"hbm_bytes_per_launch": hbm, "pixels_per_launch": px,
                   "kernel": " + ".join(sorted({l["kernel"] for l in top_levels})),
                   "source": os.path.basename(root.rstrip("/"))},
                  open(os.path.join(root, "pmc_dvo_eval.json"), "w"), indent=1)

    # the pyramid kernel (round 5: the largest single kernel of a step)
    pf = [(k, v) for k, v in fetch.items() if k[0].startswith("k_pyramid_stream")]
    pw = [(k, v) for k, v in write.items() if k[0].startswith("k_pyramid_stream")]
    if pf and pw:
        f_ = mean([x for _, v in pf for x in v["FETCH_SIZE"]])
        w_ = mean([x for _, v in pw for x in v["WRITE_SIZE"]])
        moved = (2.0 * f_ + w_) * 1024.0
        durs = [x for sub in ("trace_single",) for (n, g), d in trace_by_grid(root, sub).items()
                if n.startswith("k_pyramid_stream") for x in d]
        t = mean(durs) * 1e-9 if durs else float("nan")
        levels_px = 480 * 640 + 320 * 427 + 213 * 284
        algo = 8.0 * 256 * 3 * (480 * 640 + levels_px)     # frame read once; level 0, 1, 2 written
        print(f"\n== k_pyramid_stream (256 pairs x 3 arrays, levels 0 / 1 / 2 in one pass), alone ==")
        print(f"FETCH_SIZE raw {f_:.0f} KiB  WRITE_SIZE raw {w_:.0f} KiB -> {moved/1e9:.3f} GB moved per launch; "
              f"compulsory {algo/1e9:.3f} GB ({moved/algo:.2f}x); {t*1e6:.0f} us per launch = "
              f"{algo/t/1e9:.0f} GB/s algorithmic = {algo/t/8e12:.3f} of the 8 TB/s HBM peak")
        out["pyramid_stream"] = {"hbm_bytes_per_launch": moved, "algorithmic_bytes": algo, "avg_us_alone": t * 1e6,
                                 "frac_of_hbm_peak": algo / t / 8e12}


    for sub in ("pmc_sq", "pmc_sq2"):
        d = pmc_by_grid(root, sub)
        ev = sorted([(k, v) for k, v in d.items() if k[0].startswith("k_dvo_eval")], key=lambda kv: -kv[0][1])   # full evaluations
        if not ev:
            continue
        print(f"\n== {sub}: full-resolution k_dvo_eval, average per launch ==")
        for c, v in sorted(ev[0][1].items()):
            print(f"   {c:26s} {mean(v):18.0f}   per px {mean(v)*64/px:10.2f} (x64 lanes)")
            out.setdefault(sub, {})[c] = mean(v)
    json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
