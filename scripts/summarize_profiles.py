#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/...) into the small summaries committed under profiles/.

usage: summarize_profiles.py <tag> <kernel_stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--steps-run N]
"""
import csv
import glob
import json
import sys


def one(pattern):
    # rocprofv3 writes <dir>/<host>/<pid>_x.csv without -o and <dir>/<prefix>_x.csv with it
    f = glob.glob(pattern) or glob.glob(pattern.replace("/*/*", "/*"))
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[0]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(one(stats_dir + "/*/*_kernel_stats.csv"))))
    with open("profiles/%s_rocprof_kernel_stats.csv" % tag, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:40]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    if len(sys.argv) >= 5:
        out = {}
        for nm, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            agg = {}
            for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
                agg.setdefault(r["Kernel_Name"][:120], []).append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out.setdefault(k, {})[nm + "_KB_per_launch_max"] = max(v)
                out[k][nm + "_launches"] = len(v)
        top = dict(sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_per_launch_max", 0))[:25])
        import subprocess
        try:
            head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            head = "?"
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1; values in KB as reported by "
                           "rocprofv3 (gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams, MI355X_MICROARCH.md section HBM -- uncorrected here)",
                   "summarized_at_commit": head,
                   "kernels": top}, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)


if __name__ == "__main__":
    main()
