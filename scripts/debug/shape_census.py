"""Census of instruction shapes per level (how many distinct (opcode, term counts, coefficient ids) classes a level holds)."""
import sys, collections
sys.path.insert(0, ".")
from oracle import r1cs_format as rf
name = sys.argv[1] if len(sys.argv) > 1 else "chacha20"
cs = rf.load_reference_r1cs(name)
cd = cs.calldata
kinds = [b[0] for b in cs.blueprints]
solved = [False] * cs.nb_wires
for i in range(cs.nb_public + cs.nb_secret):
    solved[i] = True
CONST = 0xFFFFFFFF
shape_of = {}
nterms_of = {}
# instructions must be visited in order to know which wire is the unknown
for i in range(cs.nb_instructions):
    p = cs.start_calldata[i]
    size = cd[p]
    k = kinds[cs.blueprint_id[i]]
    if k == "r1c":
        n = cd[p + 1:p + 4]
        q = p + 4
        loc = 0
        sig = []
        tot = 0
        for which in range(3):
            kept = []
            for t in range(n[which]):
                c, w = cd[q], cd[q + 1]
                q += 2
                if w != CONST and not solved[w]:
                    loc = which + 1
                    solved[w] = True
                    sig.append(("u", c))
                    continue
                kept.append((c if c < 5 else "g", w == CONST))
            sig.append(tuple(kept))
            tot += len(kept)
        shape_of[i] = ("r1c", loc, tuple(sig))
        nterms_of[i] = tot
    elif k == "hint":
        hid, nin = cd[p + 1], cd[p + 2]
        o0, o1 = cd[p + size - 2], cd[p + size - 1]
        for w in range(o0, o1):
            solved[w] = True
        shape_of[i] = ("hint", hid, nin, o1 - o0, size)
        nterms_of[i] = size
    else:
        nq = cd[p + 2]
        for w in range(cs.wire_offset[i], cs.wire_offset[i] + nq):
            solved[w] = True
        shape_of[i] = ("lookup", nq, size)
        nterms_of[i] = size
print(name, "instructions", cs.nb_instructions, "levels", len(cs.levels), "distinct shapes", len(set(shape_of.values())))
tot_groups = 0
hist = collections.Counter()
for li, l in enumerate(cs.levels):
    c = collections.Counter(shape_of[i] for i in l)
    tot_groups += len(c)
    hist[len(c)] += 1
    if li < 12 or len(l) > 3000:
        print("level", li, "n", len(l), "shapes", len(c), "maxterms", max(nterms_of[i] for i in l),
              [(s[0], s[1] if s[0] != "r1c" else (s[1], [len(x) if isinstance(x, tuple) and (not x or isinstance(x[0], tuple)) else x for x in s[2]]), n) for s, n in c.most_common(6)])
print("sum over levels of shape groups:", tot_groups, "hist of groups/level:", sorted(hist.items()))
allc = collections.Counter(shape_of.values())
print("top shapes overall:")
for s, n in allc.most_common(25):
    print(n, s if len(str(s)) < 200 else str(s)[:200])
# lane-program view: wavefront tasks per level (64 instructions of a shape, or one instruction with an expression of >= 32 terms)
def is_wide(s):
    if s[0] == "r1c":
        return any(isinstance(x, tuple) and len(x) >= 32 and (not x or isinstance(x[0], tuple)) for x in s[2])
    return s[-1] >= 70
rows = []
for li, l in enumerate(cs.levels):
    c = collections.Counter(shape_of[i] for i in l)
    tasks = sum(n if is_wide(s) else (n + 63) // 64 for s, n in c.items())
    ndiv = sum(n for s, n in c.items() if s[0] == "r1c" and s[1] in (1, 2))
    rows.append((li, len(l), len(c), tasks, ndiv))
print("levels with > 16 tasks:", sum(1 for r in rows if r[3] > 16), "with divisions:", [(r[0], r[1], r[4]) for r in rows if r[4]])
print("tasks histogram:", sorted(collections.Counter(min(r[3], 100) for r in rows).items()))
