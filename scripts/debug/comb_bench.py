"""Timing of the comb quotient-MSM path alone (zkp_bench_comb) for a sweep of clump sizes.
usage: python scripts/debug/comb_bench.py LOGN BATCH K[,K...] [ITERS]"""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
logn, batch = int(sys.argv[1]), int(sys.argv[2])
ks = [int(x) for x in sys.argv[3].split(",")]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for k in ks:
    out = (C.c_double * 5)()
    rc = zk.testlib.zkp_bench_comb(logn, batch, k, iters, out)
    n = 1 << logn
    adds = ((n + k - 1) // k) * 254 * batch
    print(json.dumps({"logn": logn, "batch": batch, "k": k, "rc": rc, "build_ms": round(out[0], 1), "patterns_ms": round(out[1], 2), "walk_ms": round(out[2], 2),
                      "fold_ms": round(out[3], 2), "table_GiB": round(out[4], 2), "adds": adds, "adds_per_s": round(adds / (out[2] * 1e-3)) if out[2] > 0 else None,
                      "env_Q": os.environ.get("ZKPROVE_COMB_Q")}), flush=True)
