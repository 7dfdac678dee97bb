#!/usr/bin/env python3
"""cpu_baseline leg of bench.py: the CPU oracle (oracle/liboracle.so, C restatement of gnark's prover) timed on the
host cores.  Runs in its own interpreter (no GPU runtime) and forks one single-threaded prover per worker after
the proving key has been built, so the key is shared copy-on-write and workers do not contend on one address
space (threads in one process serialise on page-fault / allocator locks and stop scaling at ~16 workers).

usage: cpu_baseline.py <alg> <proofs_per_worker> [workers ...]   -> one JSON line (best configuration)
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_PK = None
_WITS = None


def _work(args):
    idx, n = args
    for k in range(n):
        w, inst = _WITS[(idx * n + k) % len(_WITS)]
        _PK.prove(w, inst["r"], inst["s"], inst["randomize"])
    return n


def main():
    global _PK, _WITS
    alg = int(sys.argv[1])
    per = int(sys.argv[2])
    ncpu = os.cpu_count() or 1
    quota = ncpu                 # the container may be limited by a cgroup CPU quota (the GPU boxes: 16 of 256 threads)
    try:
        q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, min(ncpu, int(round(int(q) / int(per_us)))))
    except Exception:
        pass
    workers_list = [int(x) for x in sys.argv[3:]] or sorted({quota, min(ncpu, quota + quota // 2)})
    import helpers as H
    from oracle import ref
    ref.set_threads(1)
    _PK = H.oracle_pk(alg)
    rng = H.SplitMix64(0x5EED0000 + 1)
    insts = [H.make_instance(alg, rng) for _ in range(32)]
    _WITS = [(H.oracle_witness(alg, i)[0], i) for i in insts]
    best = None
    ctx = mp.get_context("fork")
    for workers in workers_list:
        with ctx.Pool(workers) as pool:
            pool.map(_work, [(i, 0) for i in range(workers)])          # start all workers before timing
            t0 = time.perf_counter()
            done = sum(pool.map(_work, [(i, per) for i in range(workers)], chunksize=1))
            dt = time.perf_counter() - t0
        res = {"value": done / dt, "unit": "proofs/s", "cores": min(workers, quota), "kind": "port",
               "sample": "%d proofs (%d per worker) of the bench's synthetic batch in %.1f s; oracle/liboracle.so (C restatement of gnark's Groth16 "
                         "prover), one single-threaded prover process per worker, %d workers, cgroup CPU quota %d of %d host threads" % (done, per, dt, workers, quota, ncpu)}
        if best is None or res["value"] > best["value"]:
            best = res
    print(json.dumps(best), flush=True)


if __name__ == "__main__":
    main()
