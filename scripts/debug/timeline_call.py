"""usage (GPU box): python3 scripts/debug/timeline_call.py <kernel_trace.csv> [min_ms] [last_ms]
The last `last_ms` milliseconds of a kernel trace as runs of same-named kernels per hardware queue (start offset, span, count, summed time):
what a streamed ProveBatch call looks like on the device."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1200.0
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
end = max(r["e"] for r in rows)
rows = [r for r in rows if r["s"] >= end - last_ms * 1e6]
t0 = rows[0]["s"]
short = lambda n: re.sub(r"zk::|\(anonymous namespace\)::|void |Field<zk::|Params>", "", n.split("(")[0])[:48]
qid = {}
runs = {}
out = []
for r in rows:
    q = qid.setdefault(r.get("Queue_Id", "?"), len(qid))
    n = short(r["Kernel_Name"])
    cur = runs.get(q)
    if cur and cur["n"] == n and r["s"] - cur["e"] < 3e6:
        cur["e"] = max(cur["e"], r["e"]); cur["c"] += 1; cur["sum"] += r["e"] - r["s"]
    else:
        if cur: out.append(cur)
        runs[q] = {"q": q, "n": n, "s": r["s"], "e": r["e"], "c": 1, "sum": r["e"] - r["s"]}
out += [c for c in runs.values()]
out.sort(key=lambda c: c["s"])
for c in out:
    if (c["e"] - c["s"]) / 1e6 >= min_ms:
        print("q%-2d +%8.1f .. %8.1f  (%6.1f ms, x%-4d sum %6.1f)  %s" % (c["q"], (c["s"] - t0) / 1e6, (c["e"] - t0) / 1e6, (c["e"] - c["s"]) / 1e6, c["c"], c["sum"] / 1e6, c["n"]))
