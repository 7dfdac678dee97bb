"""CPU oracle throughput vs number of concurrent single-threaded provers (run on the GPU box's host)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from concurrent.futures import ThreadPoolExecutor
import helpers as H
from oracle import ref
alg = 0
pk = H.oracle_pk(alg)
rng = H.SplitMix64(1)
insts = [H.make_instance(alg, rng) for _ in range(64)]
wits = [(H.oracle_witness(alg, i)[0], i) for i in insts]
print("host threads:", os.cpu_count(), flush=True)
for omp in (1, 4):
    ref.set_threads(omp)
    for workers in (16, 32, 64, 128, 256):
        if workers * omp > 2 * (os.cpu_count() or 1): continue
        sample = (wits * (workers // len(wits) + 1))[:workers]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(lambda wi: pk.prove(wi[0], wi[1]["r"], wi[1]["s"], wi[1]["randomize"])["timings"], sample))
        dt = time.perf_counter() - t0
        print("omp=%d workers=%3d  %6.2f s  %7.2f proofs/s" % (omp, workers, dt, workers / dt), flush=True)
