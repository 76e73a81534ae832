#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short
text/JSON summary.  Usage: summarize.py <dir written by run_profile.sh>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_stats(root):
    rows = []
    for f in find(os.path.join(root, "trace"), "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    return rows


def pmc(root, sub):
    """Average counter value per dispatch, per kernel."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(os.path.join(root, sub), "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in acc.items()}


def main():
    root = sys.argv[1]
    out = {}
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for r in kernel_stats(root):
        name = r.get("Name", "")
        print(f'{name[:70]:70s} calls={r.get("Calls")} avg_ns={r.get("AverageNs")} total_ns={r.get("TotalDurationNs")} pct={r.get("Percentage")}')
        if "k_dvo_eval" in name:
            out.setdefault("kernel_stats", []).append(r)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        d = pmc(root, sub)
        print(f"== {sub} (average per dispatch) ==")
        for k, cs in d.items():
            if "k_dvo" not in k and "rescale" not in k:
                continue
            print("  ", k[:90])
            for c, (v, n) in sorted(cs.items()):
                print(f"      {c:28s} {v:18.1f}  (n={n})")
                out.setdefault(sub, {}).setdefault(k, {})[c] = v
    json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
