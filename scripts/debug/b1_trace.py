"""one synchronous ChaCha20 Prove() call at a time (what the unmodified JS operator issues), 12 calls; under rocprofv3 --kernel-trace the
last 25 ms of the trace show one call kernel by kernel (scripts/debug/timeline_call.py <trace> 0.05 25)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
zk.lib.ZkpAllowUnpinnedKeys(1)
r1cs = zk.synth.r1cs_bytes(alg)
pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rq = zk.synth.request_json(alg, zk.synth.make_instance(alg, zk.synth.SplitMix64(5)))
lat = []
for _ in range(12):
    t0 = time.perf_counter(); zk.prove_raw(rq); lat.append((time.perf_counter() - t0) * 1e3)
print("single Prove ms:", " ".join("%.2f" % x for x in lat))
