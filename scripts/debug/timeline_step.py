"""usage (GPU box): python3 scripts/debug/timeline_step.py <kernel_trace.csv> [min_ms]
One steady-state step out of a rocprofv3 kernel trace of bench.py: the window between two consecutive quotient walks (the longest
k_comb_walk launches), every kernel >= min_ms in it with its queue, start offset and duration, and per-name sums inside the window."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
walks = [r for r in rows if "k_comb_walk" in r["Kernel_Name"] and "FpParams" in r["Kernel_Name"]]
big = max(r["e"] - r["s"] for r in walks)
q = [r for r in walks if r["e"] - r["s"] > 0.6 * big]
a, b = q[len(q) - 4]["s"], q[len(q) - 3]["s"]
print("window %.2f ms between quotient walks %d and %d of %d" % ((b - a) / 1e6, len(q) - 4, len(q) - 3, len(q)))
short = lambda n: re.sub(r"zk::|\(anonymous namespace\)::|void |Field<zk::|Params>", "", n.split("(")[0])[:60]
sums = collections.Counter(); cnt = collections.Counter()
qid = {}
for r in rows:
    if r["e"] <= a or r["s"] >= b: continue
    d = (min(r["e"], b) - max(r["s"], a)) / 1e6
    n = short(r["Kernel_Name"]); sums[n] += d; cnt[n] += 1
    k = qid.setdefault(r.get("Queue_Id", "?"), len(qid))
    if d >= min_ms: print("  q%-2d +%8.2f  %7.2f ms  %s" % (k, (r["s"] - a) / 1e6, d, n))
print("sums inside the window:")
for n, v in sums.most_common(40): print("  %8.2f ms  x%-4d %s" % (v, cnt[n], n))
# busy union
ev = sorted((max(r["s"], a), min(r["e"], b)) for r in rows if r["e"] > a and r["s"] < b)
busy = 0; cur_s, cur_e = ev[0]
for s, e in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("device busy (union) %.2f ms of %.2f; sum of kernel time %.2f ms" % (busy / 1e6, (b - a) / 1e6, sum(sums.values())))
