"""usage (GPU box): python3 scripts/debug/level_times.py <kernel_trace.csv> <kernel name substring> [last_ms]
durations (us) of the individual launches of one kernel inside the last `last_ms` milliseconds of a trace, in launch order -- e.g. the
level launches of the solver in one ProveBatch call: which levels are the long ones"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
end = max(r["e"] for r in rows)
sel = [r for r in rows if r["s"] >= end - last_ms * 1e6 and name in r["Kernel_Name"]]
d = [(r["e"] - r["s"]) / 1e3 for r in sel]
gaps = [(sel[i + 1]["s"] - sel[i]["e"]) / 1e3 for i in range(len(sel) - 1)]
print("%d launches of *%s*: kernel time sum %.1f us, gaps sum %.1f us" % (len(d), name, sum(d), sum(g for g in gaps if g < 200)))
print("durations us:", " ".join("%.0f" % x for x in d))
print("gaps us:     ", " ".join("%.0f" % x for x in gaps))
print("grids:       ", " ".join("%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "")) for r in sel[:130]))
